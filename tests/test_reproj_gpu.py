"""Sparse reprojection factor: HIP keypoint reduction and the LM loop with the 5th
residual vs the oracle restatement of pvgo.py:53-61 + dense_ba.py:276-305 (oracle/reproj.py, parity unpinned: PyPose)."""
import numpy as np
import pytest
import torch

from oracle import lie, pvgo as opvgo, reproj as orp
from tests.helpers import chain_problem, reproj_inputs, se3_log_err

pytestmark = pytest.mark.gpu

LW5 = (1, 0.1, 10, 0.1, 2.0)
T_IL = np.array([0.1, -0.05, 0.02, 0.5, -0.5, 0.5, -0.5])


def _build(F, K, cuda, compat):
    from islam_amd import dense_ba, lietensor as pp
    prob, tr = chain_problem(F)
    inp = reproj_inputs(tr, K, T_IL)
    ref = orp.SparseReprojection(**inp)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    hip = dense_ba.SparseReprojectionLoss(t(inp['points2d']), t(inp['depth']), t(inp['flow']), inp['fx'], inp['fy'], inp['cx'],
                                          inp['cy'], pp.SE3(t(T_IL)), device=cuda)
    hip.compat_first_motion = compat
    return prob, tr, ref, hip


def _upper(A):
    iu = np.triu_indices(6)
    return A[..., iu[0], iu[1]]


@pytest.mark.parametrize('K,compat', [(40, True), (40, False), (300, False), (700, True)])
def test_reproj_reduce_matches_oracle(cuda, K, compat):
    """J^T J, J^T r, r^T r per link (eta coordinates) at the nodes and at a retracted trial point."""
    from islam_amd import ops
    F = 12
    prob, tr, ref, hip = _build(F, K, cuda, compat)
    nodes = prob['init_nodes']
    rng = np.random.default_rng(3)
    dx = np.concatenate([rng.normal(0, 0.02, (F, 6)), rng.normal(0, 0.1, (F, 3))], 1)
    K4 = [float(v) for v in (hip.K[0, 0], hip.K[1, 1], hip.K[0, 2], hip.K[1, 2])]
    st = ops.pvgo_reproj_struct(hip.point3d.double().contiguous(), hip.target.double().contiguous(), K4, T_IL, 1.0, compat)
    nd = torch.tensor(nodes, device=cuda)
    for d in (None, dx):
        at = nodes if d is None else lie.se3_mul(lie.se3_exp(d[:, :6]), nodes)
        m = orp.link_motions(at, compat)
        r = ref(m)                                                 # (M, K, 2)
        J = ref.jac_eta(m)                                         # (M, K, 2, 6)
        S = np.einsum('mkia,mkib->mab', J, J)
        b = np.einsum('mkia,mki->ma', J, r)
        c = (r ** 2).sum((1, 2))
        if compat:
            S[0], b[0] = 0.0, 0.0
        red = ops.pvgo_reproj_reduce(nd, st, None if d is None else torch.tensor(d, device=cuda)).cpu().numpy()
        scale = np.abs(S).max()
        np.testing.assert_allclose(red[:, :21], _upper(S), rtol=1e-10, atol=1e-12 * scale)
        np.testing.assert_allclose(red[:, 21:27], b, rtol=1e-9, atol=1e-12 * np.abs(b).max())
        np.testing.assert_allclose(red[:, 27], c, rtol=1e-11)
        assert np.all(red[:, 28:] == 0)
    # the residual evaluated through the Python surface (LieTensor ops on the device) agrees too
    from islam_amd import lietensor as pp
    e = hip(pp.SE3(torch.tensor(orp.link_motions(nodes, False), device=cuda))).cpu().numpy()
    np.testing.assert_allclose(e, ref(orp.link_motions(nodes, False)), atol=1e-9)


@pytest.mark.parametrize('F,K,compat', [(9, 40, True), (9, 40, False), (65, 150, False), (257, 64, True)])
def test_run_pvgo_with_reprojection_factor(cuda, F, K, compat):
    """run_pvgo(reproj=SparseReprojectionLoss): accept/reject trace and poses vs the oracle LM with the 5th residual."""
    from islam_amd import lietensor as pp
    from islam_amd.pvgo import run_pvgo
    prob, tr, ref, hip = _build(F, K, cuda, compat)
    out = opvgo.run_pvgo(**prob, loss_weight=LW5, mode='dense' if F <= 9 else 'banded', reproj=ref, compat_first_motion=compat,
                         return_optimizer=True)
    otl, orl, on, ov, ocov, opt = out
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    tl, rl, nodes, vels, covs, res = run_pvgo(pp.SE3(t(prob['init_nodes'])), t(prob['init_vels']),
                                              pp.SE3(t(prob['vo_motions']).to(cuda)), torch.tensor(prob['links']), t(prob['dts']),
                                              pp.SO3(t(prob['imu_drots'])), t(prob['imu_dtrans']), t(prob['imu_dvels']),
                                              device='cuda', loss_weight=LW5, reproj=hip, return_info=True)
    assert res.trials == len(opt.trace) and res.steps >= 1
    assert res.loss == pytest.approx(opt.loss, rel=1e-8)
    err = se3_log_err(nodes.numpy(), on)
    refn = np.maximum(np.linalg.norm(lie.se3_log(on), axis=-1), 1e-6)
    assert (err / refn).max() < 1e-6                      # north_star tolerance is 1e-4
    np.testing.assert_allclose(vels.numpy(), ov, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(tl.cpu().numpy(), otl, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(covs['reproj'], ocov['reproj'])
    # the factor matters: without it the optimum differs by far more than the tolerance above
    _, _, on0, _, _ = opvgo.run_pvgo(**prob, loss_weight=LW5[:4], mode='banded')
    assert se3_log_err(on0, on).max() > 1e-3


@pytest.mark.parametrize('how', ['dense', 'band_pcg'])
def test_reprojection_factor_on_a_graph_with_loop_closures(cuda, how):
    """The 5th residual couples consecutive nodes whatever `links` holds (pvgo.py:54-56), so it also works on general topologies:
    both general solvers against the oracle's dense LM with the reprojection rows (missing item 6)."""
    from islam_amd import lietensor as pp
    from islam_amd.pvgo import run_pvgo
    F, K = 21, 48
    prob, tr, ref, hip = _build(F, K, cuda, True)
    links, vo = prob['links'].copy(), prob['vo_motions'].copy()
    gt = np.concatenate([tr['gt_pos'], tr['gt_quat']], 1)
    rng = np.random.default_rng(5)
    for e, (i, j) in {3: (0, 9), 11: (4, 17), 19: (20, 2)}.items():      # replace three chain edges by long-range ones
        links[e] = (i, j)
        vo[e] = lie.se3_mul(lie.se3_mul(lie.se3_inv(gt[i]), gt[j]), lie.se3_exp(rng.normal(0, 0.01, 6)))
    p2 = dict(prob, links=links, vo_motions=vo)
    otl, orl, on, ov, ocov, opt = opvgo.run_pvgo(**p2, loss_weight=LW5, mode='dense', reproj=ref, compat_first_motion=True,
                                                 return_optimizer=True)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    tl, rl, nodes, vels, covs, res = run_pvgo(pp.SE3(t(p2['init_nodes'])), t(p2['init_vels']), pp.SE3(t(vo).to(cuda)), torch.tensor(links),
                                              t(p2['dts']), pp.SO3(t(p2['imu_drots'])), t(p2['imu_dtrans']), t(p2['imu_dvels']),
                                              device='cuda', loss_weight=LW5, reproj=hip, return_info=True, general_solver=how)
    assert res['trials'] == len(opt.trace)
    np.testing.assert_array_equal([a for _, _, a in res['trace']], [a for _, _, a in opt.trace])
    assert res['loss'] == pytest.approx(opt.loss, rel=1e-7)
    err = se3_log_err(nodes.numpy(), on)
    refn = np.maximum(np.linalg.norm(lie.se3_log(on), axis=-1), 1e-6)
    assert (err / refn).max() < 1e-6
    np.testing.assert_allclose(vels.numpy(), ov, atol=1e-7)
    np.testing.assert_allclose(covs['reproj'], ocov['reproj'])
    # the factor matters here too
    _, _, on0, _, _ = opvgo.run_pvgo(**p2, loss_weight=LW5[:4], mode='dense')
    assert se3_log_err(on0, on).max() > 1e-3
