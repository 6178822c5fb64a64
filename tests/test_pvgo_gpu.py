"""GPU parity: HIP PVGO (through the C ABI) vs the CPU oracle.  Tolerance from BASELINE.json north_star:
1e-4 relative on the SE(3) log (we assert much tighter where fp64 allows)."""
import numpy as np
import pytest
import torch

from oracle import lie, pvgo as opvgo
from tests.helpers import chain_problem, se3_log_err

pytestmark = pytest.mark.gpu
LW = (1, 0.1, 10, 0.1)


def _dev(prob, cuda):
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=cuda)
    return (t(prob['init_nodes']), t(prob['init_vels']), t(prob['vo_motions']), t(prob['imu_drots']),
            t(prob['imu_dtrans']), t(prob['imu_dvels']), t(prob['dts']))


@pytest.mark.parametrize('F', [2, 3, 9, 65, 300])
def test_linearize_and_normal_equations(cuda, F):
    from islam_amd import ops
    prob, _ = chain_problem(F)
    nodes, vels, poses, drots, dtrans, dvels, dts = _dev(prob, cuda)
    lin, part = ops.pvgo_linearize(nodes, vels, poses, drots, dtrans, dvels, dts)
    res = opvgo.residuals(prob['init_nodes'], prob['init_vels'], prob['links'], prob['vo_motions'], prob['imu_drots'],
                          prob['imu_dtrans'], prob['imu_dvels'], prob['dts'])
    A, B = opvgo.jac_blocks(prob['init_nodes'], prob['links'], prob['vo_motions'], prob['imu_drots'], res[0], res[2])
    L = lin.cpu().numpy()
    np.testing.assert_allclose(L[0:6].T, res[0], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(L[36:39].T, res[1], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(L[24:27].T, res[2], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(L[39:42].T, res[3], rtol=1e-9, atol=1e-12)
    G = L[6:15].T.reshape(-1, 3, 3)
    C = L[15:24].T.reshape(-1, 3, 3)
    np.testing.assert_allclose(G, A[:, :3, :3], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(G, A[:, 3:, 3:], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(C, A[:, :3, 3:], rtol=1e-7, atol=1e-8)
    assert np.abs(A[:, 3:, :3]).max() < 1e-12
    np.testing.assert_allclose(L[27:36].T.reshape(-1, 3, 3), B, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(part.sum().item(), opvgo.loss_unweighted(res), rtol=1e-10)
    # normal equations vs the oracle's banded assembly
    w4 = [x ** 2 for x in LW]
    Hd, Ho, rhs = ops.pvgo_build_normal(lin, dts, F, w4)
    inp = (prob['links'], prob['vo_motions'], prob['imu_drots'], prob['imu_dtrans'], prob['imu_dvels'], prob['dts'])
    bl = opvgo._BandedLin(prob['init_nodes'], inp, res, A, B, w4, False)
    N = F
    Afull = np.zeros((9 * N, 9 * N))
    for u in range(18):
        Afull[np.arange(u, 9 * N), np.arange(0, 9 * N - u)] = bl.ab[u, :9 * N - u]
    Afull = Afull + np.tril(Afull, -1).T
    np.fill_diagonal(Afull, np.clip(bl.diag, 1e-4, 1e32))
    Hd_c, Ho_c = Hd.cpu().numpy(), Ho.cpu().numpy()
    for k in range(N):
        np.testing.assert_allclose(Hd_c[k], Afull[9 * k:9 * k + 9, 9 * k:9 * k + 9], rtol=1e-8, atol=1e-8)
        if k + 1 < N:
            np.testing.assert_allclose(Ho_c[k], Afull[9 * k:9 * k + 9, 9 * k + 9:9 * k + 18], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(rhs.cpu().numpy().reshape(-1), bl.b, rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize('N,seg', [(1, (0, 0)), (2, (0, 0)), (9, (0, 0)), (40, (0, 0)), (41, (0, 0)), (64, (4, 4)),
                                   (65, (7, 4)), (100, (9, 0)), (257, (0, 0)), (1000, (0, 0)), (1000, (4, 4)),
                                   (5001, (0, 0)), (5001, (19, 15)), (5003, (24, 6)),
                                   # twisted (two-wavefront) elimination: roots of 2..7 nodes and above, pinned odd / even segment
                                   # lengths, a tree of six levels, and a chain too long for it (one-sided fallback)
                                   (13, (0, 0)), (23, (0, 0)), (47, (0, 0)), (57, (5, 5)), (64, (7, 7)), (500, (6, 5)),
                                   (5001, (7, 5)), (30011, (0, 0)), (300007, (0, 0))])
def test_block_tridiagonal_solver(cuda, N, seg):
    """Partitioned block-Cholesky vs a dense/banded CPU solve on a random SPD block-tridiagonal system."""
    from islam_amd import ops
    import scipy.linalg as sla
    rng = np.random.default_rng(N)
    # SPD by construction: A = sum over links of J^T J (J couples node k and k+1) + diagonal
    Hd = np.zeros((N, 9, 9))
    Ho = np.zeros((N, 9, 9))
    for k in range(N):
        Hd[k] += np.diag(rng.uniform(0.1, 2.0, 9))
    Jk = rng.normal(size=(max(N - 1, 0), 12, 18))
    for k in range(N - 1):
        JJ = Jk[k].T @ Jk[k]
        Hd[k] += JJ[:9, :9]
        Hd[k + 1] += JJ[9:, 9:]
        Ho[k] = JJ[:9, 9:]
    rhs = rng.normal(size=(N, 9))
    damping = 0.37
    t = lambda a: torch.tensor(a, dtype=torch.float64, device=cuda)
    Hd_d = t(Hd)
    dx = ops.pvgo_solve_chain(Hd_d, t(Ho), t(rhs), damping, seg_len=seg).cpu().numpy()
    # in-place cumulative damping of the stored diagonal (A.diagonal().add_(A.diagonal()*damping))
    dg = np.einsum('kii->ki', Hd)
    np.testing.assert_allclose(np.einsum('kii->ki', Hd_d.cpu().numpy()), dg * (1 + damping), rtol=1e-14)
    ab = np.zeros((18, 9 * N))
    for r in range(9):
        for c in range(9):
            if r >= c:
                ab[r - c, c::9] = Hd[:, r, c] * ((1 + damping) if r == c else 1.0)
            if N > 1:
                ab[9 + c - r, r:9 * (N - 1):9] = Ho[:N - 1, r, c]
    ref = sla.solveh_banded(ab, rhs.reshape(-1), lower=True).reshape(N, 9)
    scale = np.abs(ref).max()
    assert np.abs(dx - ref).max() <= 1e-9 * scale


def test_solver_reports_non_positive_definite(cuda):
    from islam_amd import ops
    from islam_amd._lib import IslamHipError
    N = 30
    Hd = torch.eye(9, dtype=torch.float64, device=cuda).repeat(N, 1, 1)
    Hd[7, 3, 3] = -1.0
    Ho = torch.zeros((N, 9, 9), dtype=torch.float64, device=cuda)
    rhs = torch.ones((N, 9), dtype=torch.float64, device=cuda)
    with pytest.raises(IslamHipError) as e:
        ops.pvgo_solve_chain(Hd, Ho, rhs, 0.0)
    assert e.value.code == -3


@pytest.mark.parametrize('F', [2, 3, 9, 12, 13, 14, 33, 63, 64, 65, 127, 257])
def test_lm_matches_oracle(cuda, F):
    """Whole LM loop: same accept/reject trace, same poses (<= 1e-4 rel on the SE3 log; here ~1e-9)."""
    from islam_amd import ops
    prob, _ = chain_problem(F)
    out = opvgo.run_pvgo(**prob, loss_weight=LW, mode='dense' if F <= 33 else 'banded', return_optimizer=True)
    opt = out[5]
    nodes, vels, poses, drots, dtrans, dvels, dts = _dev(prob, cuda)
    prm = ops.pvgo_default_params(LW, radius=1e4)
    res, trace = ops.pvgo_run_chain(nodes, vels, poses, drots, dtrans, dvels, dts, prm, trace_cap=256)
    ot = np.array([(l, d, float(a)) for l, d, a in opt.trace])
    assert res.trials == len(ot)
    np.testing.assert_array_equal(trace[:, 2], ot[:, 2])
    np.testing.assert_allclose(trace[:, 0], ot[:, 0], rtol=1e-8)
    np.testing.assert_allclose(trace[:, 1], ot[:, 1], rtol=1e-12)
    err = se3_log_err(nodes.cpu().numpy(), opt.nodes)
    ref = np.maximum(np.linalg.norm(lie.se3_log(opt.nodes), axis=-1), 1e-6)
    assert (err / ref).max() < 1e-6
    np.testing.assert_allclose(vels.cpu().numpy(), opt.vels, rtol=1e-6, atol=1e-8)


def test_lm_full_size_properties(cuda):
    """BASELINE config 4 size (N=5001): oracle (banded) parity + size-independent properties."""
    from islam_amd import ops
    prob, _ = chain_problem(5001)
    out = opvgo.run_pvgo(**prob, loss_weight=LW, mode='banded', return_optimizer=True)
    opt = out[5]
    nodes, vels, poses, drots, dtrans, dvels, dts = _dev(prob, cuda)
    n0 = nodes.clone()
    prm = ops.pvgo_default_params(LW, radius=1e4)
    res, trace = ops.pvgo_run_chain(nodes, vels, poses, drots, dtrans, dvels, dts, prm, trace_cap=256)
    ot = np.array([(l, d, float(a)) for l, d, a in opt.trace])
    assert res.trials == len(ot)
    np.testing.assert_array_equal(trace[:, 2], ot[:, 2])
    err = se3_log_err(nodes.cpu().numpy(), opt.nodes)
    ref = np.maximum(np.linalg.norm(lie.se3_log(opt.nodes), axis=-1), 1e-6)
    assert (err / ref).max() < 1e-4
    # quaternions stay unit, accepted losses are monotone, the run is deterministic
    q = nodes[:, 3:].cpu().numpy()
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-9)
    acc = trace[trace[:, 2] == 1.0, 0]
    nodes2, vels2 = n0.clone(), torch.tensor(prob['init_vels'], dtype=torch.float64, device=cuda)
    res2, trace2 = ops.pvgo_run_chain(nodes2, vels2, poses, drots, dtrans, dvels, dts, prm, trace_cap=256)
    np.testing.assert_array_equal(trace, trace2)
    assert torch.equal(nodes, nodes2)
    assert len(acc) >= 1


def test_vo_loss_and_align(cuda):
    from islam_amd import ops
    prob, _ = chain_problem(40)
    nodes = torch.tensor(prob['init_nodes'], dtype=torch.float64, device=cuda)
    vels = torch.tensor(prob['init_vels'], dtype=torch.float64, device=cuda)
    edges = torch.tensor(prob['links'], dtype=torch.int64, device=cuda)
    poses = torch.tensor(prob['vo_motions'], dtype=torch.float64, device=cuda, requires_grad=True)
    tl, rl = ops.pvgo_vo_loss(nodes, edges, poses)
    otl, orl, _ = opvgo.vo_loss(prob['init_nodes'], prob['links'], prob['vo_motions'])
    np.testing.assert_allclose(tl.detach().cpu().numpy(), otl, rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(rl.detach().cpu().numpy(), orl, rtol=1e-9, atol=1e-14)
    gt = np.linspace(0.5, 1.5, 39)
    gr = np.linspace(2.0, 1.0, 39)
    (tl * torch.tensor(gt, device=cuda) + rl * torch.tensor(gr, device=cuda)).sum().backward()
    og = opvgo.vo_loss_grad(prob['init_nodes'], prob['links'], prob['vo_motions'], gt, gr)
    np.testing.assert_allclose(poses.grad.cpu().numpy(), og, rtol=1e-8, atol=1e-12)
    tgt = np.array([1.0, -2.0, 0.5, 0.1, 0.2, -0.1, 0.0])
    tgt[6] = np.sqrt(1 - np.sum(tgt[3:6] ** 2))
    an, av = ops.pvgo_align(nodes, vels, torch.tensor(tgt, dtype=torch.float64, device=cuda))
    on, ov = opvgo.align_to(prob['init_nodes'], prob['init_vels'], tgt)
    np.testing.assert_allclose(an.cpu().numpy(), on, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(av.cpu().numpy(), ov, rtol=1e-10, atol=1e-12)


def _noisy_problem(F, seed, sig):
    """Dead-reckoning init perturbed hard enough that LM has to reject trials (the run-ahead / epoch-gate paths)."""
    prob, _ = chain_problem(F)
    rng = np.random.default_rng(seed)
    n = prob['init_nodes'].copy()
    n[:, :3] += rng.normal(0, sig, (F, 3))
    n = lie.se3_mul(lie.se3_exp(np.concatenate([np.zeros((F, 3)), rng.normal(0, sig * 0.2, (F, 3))], 1)), n)
    return dict(prob, init_nodes=n)


@pytest.mark.parametrize('F,seed,sig,pattern', [(33, 1, 1.5, '00000011110'), (65, 8, 1.5, '0000000010'),
                                               (65, 2, 1.0, None)])
def test_lm_with_rejected_trials_matches_oracle(cuda, F, seed, sig, pattern):
    """Rejected trials (more damping on the same linearisation, the pre-enqueued run-ahead iteration cancelled by the
    epoch gate) and the reject limit (16 rejects, then the step is kept and the scheduler stops): same accept / reject
    sequence, losses, dampings and final iterate as the oracle."""
    from islam_amd import ops
    prob = _noisy_problem(F, seed, sig)
    out = opvgo.run_pvgo(**prob, loss_weight=LW, mode='banded', return_optimizer=True)
    opt = out[5]
    rej = ''.join(str(int(not t[2])) for t in opt.trace)
    if pattern is not None:
        assert rej == pattern, 'the oracle problem changed: %s' % rej
    else:
        assert rej.count('1') >= 16                    # reject limit reached
    nodes, vels, poses, drots, dtrans, dvels, dts = _dev(prob, cuda)
    prm = ops.pvgo_default_params(LW, radius=1e4)
    res, trace = ops.pvgo_run_chain(nodes, vels, poses, drots, dtrans, dvels, dts, prm, trace_cap=256)
    ot = np.array([(l, d, float(a)) for l, d, a in opt.trace])
    assert res.trials == len(ot)
    np.testing.assert_array_equal(trace[:, 2], ot[:, 2])
    np.testing.assert_allclose(trace[:, 0], ot[:, 0], rtol=1e-7)
    np.testing.assert_allclose(trace[:, 1], ot[:, 1], rtol=1e-12)
    assert res.loss == pytest.approx(opt.loss, rel=1e-7)
    err = se3_log_err(nodes.cpu().numpy(), opt.nodes)
    ref = np.maximum(np.linalg.norm(lie.se3_log(opt.nodes), axis=-1), 1e-6)
    assert (err / ref).max() < 1e-6
    np.testing.assert_allclose(vels.cpu().numpy(), opt.vels, rtol=1e-6, atol=1e-7)
    # a second run on the same workspace (stale ready words, cancelled kernels still draining) gives the same answer
    n2, v2 = _dev(prob, cuda)[:2]
    res2, _ = ops.pvgo_run_chain(n2, v2, poses, drots, dtrans, dvels, dts, prm)
    assert res2.trials == res.trials and res2.steps == res.steps
    torch.testing.assert_close(n2, nodes, rtol=0, atol=0)


def test_lm_solver_failure_breaks_the_step_like_pypose(cuda):
    """An indefinite normal matrix (a negative information scalar) makes the Cholesky factorisation fail: PyPose prints
    "Linear solver failed. Breaking optimization step..." and keeps looping through the scheduler until the plateau counter
    stops it (3 steps, one failed solve each); the iterate must not move."""
    from islam_amd import ops
    prob, _ = chain_problem(33)
    info = (1.0, -0.5, 100.0, 0.01)
    out = opvgo.run_pvgo(**prob, loss_weight=LW, mode='banded', return_optimizer=True, info_scalars=info)
    opt = out[5]
    assert len(opt.trace) == 0 and np.array_equal(opt.nodes, prob['init_nodes'])      # every solve failed, nothing moved
    nodes, vels, poses, drots, dtrans, dvels, dts = _dev(prob, cuda)
    n0 = nodes.clone()
    prm = ops.pvgo_default_params(LW, radius=1e4)
    for i in range(4):
        prm.w[i] = info[i]
    res, _ = ops.pvgo_run_chain(nodes, vels, poses, drots, dtrans, dvels, dts, prm)
    assert res.status == -3                                     # ISLAM_ENOTPD
    assert res.steps == 3 and res.trials == 3                   # StopOnPlateau(patience=3): no decrease three times
    assert res.loss == pytest.approx(opt.loss, rel=1e-9)
    torch.testing.assert_close(nodes, n0, rtol=0, atol=0)
    # the workspace is reusable afterwards
    prm2 = ops.pvgo_default_params(LW, radius=1e4)
    res2, _ = ops.pvgo_run_chain(nodes, vels, poses, drots, dtrans, dvels, dts, prm2)
    ref = opvgo.run_pvgo(**prob, loss_weight=LW, mode='banded', return_optimizer=True)[5]
    assert res2.status == 0 and res2.trials == len(ref.trace)


def test_eliminate_level0_hook_is_the_solvers_first_launch(cuda):
    """islam_pvgo_eliminate_level0 (bench.py's roofline leg) launches exactly what a solve launches first: it damps the
    stored diagonal like a solve does, writes nothing the caller owns, and rejects single-level problems."""
    from islam_amd import ops
    from islam_amd._lib import IslamHipError, c_double, c_int, c_size_t, check, lib, ptr, stream_ptr
    N = 700
    g = torch.Generator().manual_seed(1)
    Hd = torch.eye(9, dtype=torch.float64).repeat(N, 1, 1) * 30 + 0.1 * torch.randn(N, 9, 9, generator=g, dtype=torch.float64)
    Hd = (Hd + Hd.transpose(1, 2)).contiguous().to(cuda)
    Ho = (0.2 * torch.randn(N, 9, 9, generator=g, dtype=torch.float64)).to(cuda)
    rhs = torch.randn(N, 9, generator=g, dtype=torch.float64).to(cuda)
    ws, nbytes = ops.pvgo_workspace(N, cuda)
    sl = (c_int * 2)(0, 0)
    Hd2, Ho2, rhs2 = Hd.clone(), Ho.clone(), rhs.clone()
    check(lib().islam_pvgo_eliminate_level0(ptr(Hd2), ptr(Ho2), ptr(rhs2), c_double(0.25), N, sl, ptr(ws), c_size_t(nbytes),
                                            stream_ptr(cuda)))
    torch.cuda.synchronize()
    dg = torch.diagonal(Hd, dim1=1, dim2=2)
    torch.testing.assert_close(torch.diagonal(Hd2, dim1=1, dim2=2), dg * 1.25, rtol=1e-14, atol=0)
    off = ~torch.eye(9, dtype=torch.bool, device=cuda)
    assert torch.equal(Hd2[:, off], Hd[:, off]) and torch.equal(Ho2, Ho) and torch.equal(rhs2, rhs)
    # the workspace it left behind is a valid start for a full solve (nothing persistent was corrupted)
    dx = ops.pvgo_solve_chain(Hd.clone(), Ho, rhs, 0.25, workspace=(ws, nbytes))
    dx_ref = ops.pvgo_solve_chain(Hd.clone(), Ho, rhs, 0.25)
    assert torch.equal(dx, dx_ref)
    with pytest.raises(IslamHipError):
        check(lib().islam_pvgo_eliminate_level0(ptr(Hd2), ptr(Ho2), ptr(rhs2), c_double(0.0), 5, sl, ptr(ws), c_size_t(nbytes),
                                                stream_ptr(cuda)))


def test_block_tridiagonal_solver_random_sizes_and_segment_lengths(cuda):
    """40 random chains (2..1500 nodes) with random pinned segment lengths: twisted and one-sided plans, every root size."""
    from islam_amd import ops
    import scipy.linalg as sla
    rng0 = np.random.default_rng(777)
    for case in range(40):
        N = int(rng0.integers(2, 1500))
        seg = (0, 0) if case % 3 == 0 else (int(rng0.integers(0, 9)), int(rng0.integers(0, 9)))
        rng = np.random.default_rng(1000 + case)
        Hd = np.zeros((N, 9, 9))
        Ho = np.zeros((N, 9, 9))
        Hd[:, np.arange(9), np.arange(9)] = rng.uniform(0.1, 2.0, (N, 9))
        JJ = np.einsum('kri,krj->kij', *(2 * [rng.normal(size=(N - 1, 12, 18))]))
        Hd[:-1] += JJ[:, :9, :9]
        Hd[1:] += JJ[:, 9:, 9:]
        Ho[:N - 1] = JJ[:, :9, 9:]
        rhs = rng.normal(size=(N, 9))
        damping = float(rng.uniform(0, 1))
        t = lambda a: torch.tensor(a, dtype=torch.float64, device=cuda)
        dx = ops.pvgo_solve_chain(t(Hd), t(Ho), t(rhs), damping, seg_len=seg).cpu().numpy()
        ab = np.zeros((18, 9 * N))
        for r in range(9):
            for c in range(9):
                if r >= c:
                    ab[r - c, c::9] = Hd[:, r, c] * ((1 + damping) if r == c else 1.0)
                ab[9 + c - r, r:9 * (N - 1):9] = Ho[:N - 1, r, c]
        ref = sla.solveh_banded(ab, rhs.reshape(-1), lower=True).reshape(N, 9)
        assert np.abs(dx - ref).max() <= 1e-9 * np.abs(ref).max(), (case, N, seg)


def test_rejecting_bench_configuration_matches_oracle(cuda):
    """bench.py's `reject_heavy` leg: the N = 5001 graph from its dead-reckoning start under TrustRegion(radius=1e8).  The first
    optimizer.step rejects its trial four times (damping 2e-8 -> 8e-8 -> 6.4e-7 -> 1.02e-5) before it accepts: same accept / reject
    sequence, dampings, losses and final iterate as the oracle (pvgo.py:168-180 with radius 1e8 instead of 1e4)."""
    from islam_amd import ops
    prob, _ = chain_problem(5001)
    out = opvgo.run_pvgo(**prob, loss_weight=LW, mode='banded', return_optimizer=True, radius=1e8)
    opt = out[5]
    rej = ''.join(str(int(not t[2])) for t in opt.trace)
    assert rej == '11110000000000', 'the oracle problem changed: %s' % rej
    nodes, vels, poses, drots, dtrans, dvels, dts = _dev(prob, cuda)
    res, trace = ops.pvgo_run_chain(nodes, vels, poses, drots, dtrans, dvels, dts, ops.pvgo_default_params(LW, radius=1e8), trace_cap=256)
    ot = np.array([(l, d, float(a)) for l, d, a in opt.trace])
    assert res.trials == len(ot)
    np.testing.assert_array_equal(trace[:, 2], ot[:, 2])
    np.testing.assert_allclose(trace[:, 1], ot[:, 1], rtol=1e-12)
    np.testing.assert_allclose(trace[:, 0], ot[:, 0], rtol=1e-6)
    err = se3_log_err(nodes.cpu().numpy(), opt.nodes)
    ref = np.maximum(np.linalg.norm(lie.se3_log(opt.nodes), axis=-1), 1e-6)
    assert (err / ref).max() < 1e-4                    # north_star's bound on the SE(3) log


@pytest.mark.parametrize('F,seed,sig', [(1000, 3, 0.8), (5001, 5, 0.5), (5001, 6, 1.5), (777, 9, 3.0)])
def test_fused_loop_equals_the_launch_per_stage_loop_on_reject_heavy_graphs(cuda, F, seed, sig, monkeypatch):
    """islam_pvgo_run_chain's two loops on problems that make LM reject trials and change its damping at full size: the fused loop
    (trial_elim_kernel: speculated damping, verdict 5 / reject / fallback solves from the undamped linearisation + damping history)
    against the launch-per-stage loop (in-place cumulative damping) -- same accept / reject sequence, same dampings, same iterate."""
    from islam_amd import ops
    prob = _noisy_problem(F, seed, sig)
    prm = ops.pvgo_default_params(LW, radius=1e4)
    outs = []
    for no_fuse in ('1', '0'):
        monkeypatch.setenv('ISLAM_PVGO_NO_FUSE', no_fuse)
        nodes, vels, poses, drots, dtrans, dvels, dts = _dev(prob, cuda)
        res, trace = ops.pvgo_run_chain(nodes, vels, poses, drots, dtrans, dvels, dts, prm, trace_cap=256)
        outs.append((res, np.asarray(trace)[:res.trials], nodes.cpu().numpy(), vels.cpu().numpy()))
    (r0, t0, n0, v0), (r1, t1, n1, v1) = outs
    assert (r1.trials, r1.steps, r1.status) == (r0.trials, r0.steps, r0.status)
    np.testing.assert_array_equal(t1[:, 2], t0[:, 2])                       # accept / reject pattern
    np.testing.assert_allclose(t1[:, 1], t0[:, 1], rtol=1e-12)              # dampings
    np.testing.assert_allclose(t1[:, 0], t0[:, 0], rtol=1e-9)               # trial losses
    np.testing.assert_allclose(n1, n0, rtol=0, atol=1e-8)
    np.testing.assert_allclose(v1, v0, rtol=0, atol=1e-8)
    if sig >= 1.5:
        assert r0.trials > r0.steps or len(set(np.round(t0[:, 1] / t0[0, 1], 6))) > 1      # rejects or a changing damping were exercised
