"""GPU tests of the drop-in Python surface (run_pvgo / IMUModule / TartanVO / bilevel loop) against the oracle."""
import numpy as np
import pytest
import torch

from islam_amd import synthetic
from oracle import imu as oimu, lie, pvgo as opvgo
from tests.helpers import chain_problem, se3_log_err

pytestmark = pytest.mark.gpu
LW = (1, 0.1, 10, 0.1)


def test_run_pvgo_surface_matches_oracle(cuda):
    """train.py-shaped call: float32 CPU inputs, VO motions on the device with grad; outputs as pvgo.py:205."""
    from islam_amd import lietensor as pp
    from islam_amd.pvgo import run_pvgo
    prob, _ = chain_problem(9)
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
    vo = f32(prob['vo_motions']).to(cuda).requires_grad_(True)
    tl, rl, nodes, vels, covs = run_pvgo(pp.SE3(f32(prob['init_nodes'])), f32(prob['init_vels']), pp.SE3(vo),
                                         torch.tensor(prob['links']), f32(prob['dts']), pp.SO3(f32(prob['imu_drots'])),
                                         f32(prob['imu_dtrans']), f32(prob['imu_dvels']), device='cuda', radius=1e4,
                                         loss_weight=LW, target='vo')
    p32 = {k: (np.asarray(v, np.float32).astype(np.float64) if k != 'links' else v) for k, v in prob.items()}
    otl, orl, on, ov, ocov = opvgo.run_pvgo(**p32, loss_weight=LW, mode='dense')
    assert nodes.device.type == 'cpu' and vels.device.type == 'cpu' and nodes.dtype == torch.float32
    assert isinstance(nodes, pp.LieTensor) and nodes.shape == (9, 7) and vels.shape == (9, 3)
    err = se3_log_err(nodes.numpy().astype(np.float64), on)
    ref = np.maximum(np.linalg.norm(lie.se3_log(on), axis=-1), 1e-6)
    assert (err / ref).max() < 1e-4                                   # north_star tolerance (float32 I/O rounding included)
    np.testing.assert_allclose(vels.numpy(), ov, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tl.detach().cpu().numpy(), otl, rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(rl.detach().cpu().numpy(), orl, rtol=2e-3, atol=1e-9)
    assert set(covs) == set(ocov) and all(np.array_equal(covs[k], ocov[k]) for k in covs)
    # one-step back-propagation (train.py:280-283): gradient reaches the VO motions, PyPose convention (7th entry 0)
    loss_bp = torch.cat((1.0 * rl, 0.1 * tl))
    loss_bp.backward(torch.ones_like(loss_bp))
    g = vo.grad.cpu().numpy()
    og = opvgo.vo_loss_grad(opvgo.run_pvgo(**p32, loss_weight=LW, mode='dense', return_optimizer=True)[5].nodes, prob['links'],
                            p32['vo_motions'], np.full(8, 0.1), np.full(8, 1.0))
    assert np.all(g[:, 6] == 0)
    np.testing.assert_allclose(g, og, rtol=5e-3, atol=1e-5)


def test_run_pvgo_rejects_what_is_not_built(cuda):
    from islam_amd.pvgo import run_pvgo
    prob, _ = chain_problem(5)
    args = [torch.tensor(prob[k]) for k in ('init_nodes', 'init_vels', 'vo_motions')]
    rest = [torch.tensor(prob[k]) for k in ('dts', 'imu_drots', 'imu_dtrans', 'imu_dvels')]
    with pytest.raises(AttributeError):          # like the reference: a reproj object without .N fails at pvgo.py:131
        run_pvgo(*args, torch.tensor(prob['links']), *rest, device='cuda', loss_weight=(1, 1, 1, 1, 1), reproj=object())
    with pytest.raises(RuntimeError):
        run_pvgo(*args, torch.tensor(prob['links']), *rest, device='cpu')


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_imu_module_matches_oracle(cuda, dtype):
    from islam_amd.imu_integrator import IMUModule
    tr = synthetic.car_trajectory(41, seed=5)
    bias_a, bias_g = np.array([0.01, -0.02, 0.03]), np.array([1e-3, 2e-3, -1e-3])
    mod = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], bias_a, bias_g, tr['init'], tr['gravity'], tr['rgb2imu_sync'],
                    device='cuda', denoise_model_name=None, denoise_accel=True, denoise_gyro=False, dtype=dtype)
    npdt = {torch.float32: np.float32, torch.float64: np.float64}[dtype]
    init = dict(pos=tr['gt_pos'][8], rot=tr['gt_quat'][8], vel=tr['gt_vel'][8])
    for motion in (False, True):
        pos, rot, covs, vel = mod.integrate(8, 16, init, motion_mode=motion)
        ref = oimu.integrate(tr['accels'], tr['gyros'], tr['imu_dts'], tr['rgb2imu_sync'], 8, 16, init, tr['gravity'], motion,
                             accel_bias=bias_a, gyro_bias=None, dtype=npdt)
        assert covs == [] and pos.device.type == 'cpu' and pos.shape[0] == (8 if motion else 9)
        np.testing.assert_array_equal(pos.numpy(), ref[0])
        np.testing.assert_array_equal(rot.tensor().numpy(), ref[1])
        np.testing.assert_array_equal(vel.numpy(), ref[2])


def test_tartanvo_forward_and_bilevel_step(cuda):
    """Plumbing of BASELINE config 1/2 at reduced batch: TartanVO forward (HIP correlation/warp/scale, train-mode BN),
    IMU, PVGO, backward into the pose head only; then one optimizer step."""
    from islam_amd import lietensor as pp
    from islam_amd.TartanVO import TartanVO
    from islam_amd.bilevel import BilevelLoop
    from islam_amd.imu_integrator import IMUModule
    torch.manual_seed(0)
    B = 2
    vo = TartanVO(correct_scale=False, fix_parts=('flow', 'stereo'), use_kitti_coord=True)
    # random weights predict garbage disparity; pin the stereo head to a constant 10 px so the scale mask is non-empty
    with torch.no_grad():
        vo.vonet.stereoNet.conv_c13.weight.zero_()
        vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
    tr = synthetic.car_trajectory(2 * B + 1, seed=9)
    imu = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], np.zeros(3), np.zeros(3), tr['init'], tr['gravity'],
                    tr['rgb2imu_sync'], device='cuda', denoise_model_name=None, denoise_accel=True, denoise_gyro=False)
    loop = BilevelLoop(vo, imu, pp.identity_SE3(), tr['init'], batch_size=B)
    w0 = [p.detach().clone() for p in vo.vonet.flowPoseNet.parameters()]
    for k in range(2):
        sample = synthetic.stereo_batch(B, seed=100 + k)
        sample['link'] = sample['link'] + k * B
        loss = loop.step(sample)
        assert np.isfinite(loss)
    res = vo(synthetic.stereo_batch(B, seed=7))
    assert res['motion'].shape == (B, 7) and res['flow'].shape == (B, 2, 112, 160) and res['disp'].shape == (B, 1, 112, 160)
    assert res['mask'].dtype == torch.bool and res['depth'].shape == (B, 112, 160)
    q = res['motion'].tensor()[:, 3:]
    torch.testing.assert_close(q.norm(dim=1), torch.ones(B, device=q.device), rtol=1e-4, atol=1e-4)
    grads = [p.grad for p in vo.vonet.flowPoseNet.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads) and any(g.abs().sum() > 0 for g in grads)
    assert all(p.grad is None for p in vo.vonet.flowNet.parameters())          # frozen (F5)
    loop.end_epoch()
    assert any(not torch.equal(a, b.detach()) for a, b in zip(w0, vo.vonet.flowPoseNet.parameters()))
    assert len(loop.pgo_poses) == 2 * B + 1


def test_stereo_scale_gradient_matches_autograd(cuda):
    """d scale / d pose from the kernel's first-order sums vs plain autograd through the reference formula."""
    from islam_amd import lietensor as pp
    from islam_amd.TartanVO import stereo_scale
    B, H, W = 2, 40, 56
    g = torch.Generator().manual_seed(1)
    disp = (torch.rand(B, 1, H, W, generator=g) * 10 + 2).to(cuda)
    flow = (torch.randn(B, 2, H, W, generator=g) * 2).to(cuda)
    intr = torch.tensor([[60.0, 60.0, 28.0, 20.0]]).repeat(B, 1)
    base = torch.tensor([0.5, 0.3])
    th = torch.tensor([1.0, 1.0])
    xi = (torch.randn(B, 6, generator=g, dtype=torch.float64) * 0.2).to(cuda).requires_grad_(True)
    pose = pp.se3(xi).Exp()
    s, z, mask, dmask = stereo_scale(disp, flow, pose, intr, base, None, th)
    s.sum().backward()
    g_kernel = xi.grad.clone()
    # reference formula with torch autograd on the same masked pixels (dense_ba.py:135-166), float64
    xi2 = xi.detach().clone().requires_grad_(True)
    Tinv = pp.se3(xi2).Exp().Inv()
    tot = 0
    for b in range(B):
        fx, fy, cx, cy = [float(v) for v in intr[b]]
        u, v = torch.meshgrid(torch.arange(W, dtype=torch.float64, device=cuda), torch.arange(H, dtype=torch.float64, device=cuda), indexing='xy')
        zz = z[b].double()
        P = torch.stack([zz * (u - cx) / fx, zz * (v - cy) / fy, zz], -1)
        R, t = Tinv[b].rotation(), Tinv[b].translation()
        tn = torch.nn.functional.normalize(t, dim=0)
        K = torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=torch.float64, device=cuda)
        a = K @ tn
        bb = (R.Act(P.reshape(-1, 3)) @ K.T).reshape(H, W, 3)
        fu, fv = flow[b, 0].double() + u, flow[b, 1].double() + v
        M1, w1 = a[2] * fu - a[0], bb[..., 0] - bb[..., 2] * fu
        M2, w2 = a[2] * fv - a[1], bb[..., 1] - bb[..., 2] * fv
        m = mask[b]
        tot = tot + (M1[m] * w1[m] + M2[m] * w2[m]).sum() / (M1[m] ** 2 + M2[m] ** 2).sum()
    tot.backward()
    torch.testing.assert_close(g_kernel, xi2.grad, rtol=2e-3, atol=1e-5)


def test_flow_net_trains_end_to_end(cuda):
    """with correlation backward + warp backward the flow network can be un-frozen."""
    from islam_amd import nets
    torch.manual_seed(0)
    net = nets.PWCDCNet().to(cuda)
    x = torch.rand(1, 6, 128, 192, device=cuda)
    flows, _ = net(x)
    sum(f.abs().mean() for f in flows).backward()
    for name in ('conv1a.0.weight', 'conv6b.0.weight', 'conv3_0.0.weight', 'upfeat4.weight', 'dc_conv7.weight'):
        g = dict(net.named_parameters())[name].grad
        assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0, name


def test_run_pvgo_general_topology_matches_oracle(cuda):
    """Loop closures / non-consecutive links: dense path on the GPU vs the oracle's dense LM."""
    from islam_amd import lietensor as pp
    from islam_amd.pvgo import run_pvgo
    F = 21
    prob, tr = chain_problem(F)
    links = prob['links'].copy()
    vo = prob['vo_motions'].copy()
    gt = np.concatenate([tr['gt_pos'], tr['gt_quat']], 1)
    rng = np.random.default_rng(5)
    for e, (i, j) in {3: (0, 9), 11: (4, 17), 19: (20, 2)}.items():      # replace three chain edges by long-range ones
        links[e] = (i, j)
        rel = lie.se3_mul(lie.se3_inv(gt[i]), gt[j])
        vo[e] = lie.se3_mul(rel, lie.se3_exp(rng.normal(0, 0.01, 6)))
    p2 = dict(prob, links=links, vo_motions=vo)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    tl, rl, nodes, vels, _ = run_pvgo(pp.SE3(t(p2['init_nodes'])), t(p2['init_vels']), pp.SE3(t(vo).to(cuda)), torch.tensor(links),
                                      t(p2['dts']), pp.SO3(t(p2['imu_drots'])), t(p2['imu_dtrans']), t(p2['imu_dvels']),
                                      device='cuda', loss_weight=LW)
    otl, orl, on, ov, _ = opvgo.run_pvgo(**p2, loss_weight=LW, mode='dense')
    err = se3_log_err(nodes.numpy(), on)
    ref = np.maximum(np.linalg.norm(lie.se3_log(on), axis=-1), 1e-6)
    assert (err / ref).max() < 1e-6
    np.testing.assert_allclose(vels.numpy(), ov, atol=1e-7)
    np.testing.assert_allclose(tl.cpu().numpy(), otl, rtol=1e-6, atol=1e-10)


def test_dense_assembly_matches_dense_jacobian_product(cuda):
    """islam_pvgo_assemble_dense (block assembly, no J) == J^T W J / -J^T W r of the oracle's dense Jacobian, node-major."""
    from islam_amd import ops, pvgo_dense
    from islam_amd._lib import c_double, check, lib, ptr, stream_ptr
    F = 17
    prob, tr = chain_problem(F)
    links = prob['links'].copy()
    links[2], links[9], links[13] = (0, 7), (12, 3), (16, 1)
    links[5] = (5, 6)
    N, E, M = F, F - 1, F - 1
    res = opvgo.residuals(prob['init_nodes'], prob['init_vels'], links, prob['vo_motions'], prob['imu_drots'], prob['imu_dtrans'],
                          prob['imu_dvels'], prob['dts'])
    Ae, Bk = opvgo.jac_blocks(prob['init_nodes'], links, prob['vo_motions'], prob['imu_drots'], res[0], res[2])
    J10 = opvgo.jacobian_dense(N, links, Ae, Bk, prob['dts'])
    cols = np.concatenate([np.concatenate([7 * k + np.arange(6), 7 * N + 3 * k + np.arange(3)]) for k in range(N)])
    J = J10[:, cols]
    w = opvgo.weight_vector(E, M, LW, np.float64)
    A_ref = (J.T * w) @ J
    b_ref = -(J.T * w) @ np.concatenate([r.reshape(-1) for r in res])

    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=cuda)
    nodes, vels, poses = t(prob['init_nodes']), t(prob['init_vels']), t(prob['vo_motions'])
    drots, dtrans, dvels, dts = t(prob['imu_drots']), t(prob['imu_dtrans']), t(prob['imu_dvels']), t(prob['dts'])
    edges = torch.tensor(links, device=cuda)
    dummy = torch.zeros((M, 7), dtype=torch.float64, device=cuda)
    dummy[:, 6] = 1.0
    vo, lin = pvgo_dense._linearize(nodes, vels, edges, poses, drots, dtrans, dvels, dts, dummy)
    w2 = [x ** 2 for x in LW]
    Hd, Ho, rhs = ops.pvgo_build_normal(lin, dts, N, (0.0, w2[1], w2[2], w2[3]), -1e300, 1e300)
    nptr, nadj = [torch.from_numpy(a).to(cuda) for a in pvgo_dense._node_adjacency(links, N)]
    A = torch.full((9 * N, 9 * N), float('nan'), dtype=torch.float64, device=cuda)
    b = torch.empty(9 * N, dtype=torch.float64, device=cuda)
    check(lib().islam_pvgo_assemble_dense(ptr(Hd), ptr(Ho), ptr(rhs), ptr(vo), ptr(edges), ptr(nptr), ptr(nadj), c_double(w2[0]), N, E, ptr(A), ptr(b),
                                          stream_ptr(cuda)))
    scale = np.abs(A_ref).max()
    np.testing.assert_allclose(A.cpu().numpy(), A_ref, atol=1e-12 * scale)
    np.testing.assert_allclose(b.cpu().numpy(), b_ref, atol=1e-11 * np.abs(b_ref).max())
    D = t(np.random.default_rng(0).normal(size=(N, 9)) * 0.01)
    q = float(pvgo_dense._quality_term(vo, lin, edges, dts, D))
    D10 = np.zeros(10 * N)
    D10[cols] = D.cpu().numpy().reshape(-1)
    JD = J10 @ D10
    R = np.concatenate([r.reshape(-1) for r in res])
    assert q == pytest.approx(float(JD @ (2 * R + JD)), rel=1e-10)


def test_host_glue_matches_device_glue(cuda):
    """TartanVO(host_glue=True): the post-network pose algebra in float64 on the host gives the same motions and the same
    gradients to the pose head as the per-op device path."""
    from islam_amd.TartanVO import TartanVO
    outs = []
    sample = synthetic.stereo_batch(2, seed=3)
    for host in (False, True):
        torch.manual_seed(0)
        vo = TartanVO(correct_scale=False, fix_parts=('flow', 'stereo'), use_kitti_coord=True, host_glue=host)
        with torch.no_grad():
            vo.vonet.stereoNet.conv_c13.weight.zero_()
            vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
        res = vo(sample)
        m = res['motion']
        assert m.tensor().is_cuda and m.tensor().dtype == torch.float32
        w = torch.linspace(0.5, 1.5, 7, device=m.tensor().device)
        (res.get('motion_host', m).tensor().to(m.tensor().device).float() * w).sum().backward()
        grads = torch.cat([p.grad.reshape(-1) for p in vo.vonet.flowPoseNet.parameters() if p.grad is not None])
        outs.append((m.tensor().detach().clone(), grads.clone(), 'motion_host' in res))
    assert outs[1][2] and not outs[0][2]
    torch.testing.assert_close(outs[1][0], outs[0][0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(outs[1][1], outs[0][1], rtol=2e-3, atol=1e-6 + 1e-3 * float(outs[0][1].abs().max()))


def test_prefetched_frozen_forward_is_identical(cuda):
    """TartanVO.prefetch (frozen nets of the next batch on a side stream) changes the schedule, not the result."""
    from islam_amd.TartanVO import TartanVO
    torch.manual_seed(0)
    vo = TartanVO(correct_scale=False, fix_parts=('flow', 'stereo'), use_kitti_coord=True)
    with torch.no_grad():
        vo.vonet.stereoNet.conv_c13.weight.zero_()
        vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
    vo.vonet.stereoNet.eval()                       # keep BatchNorm statistics fixed so both passes see the same network
    a, b = synthetic.stereo_batch(2, seed=3), synthetic.stereo_batch(2, seed=4)
    ref = [vo(s, is_train=False)['motion'].tensor().clone() for s in (a, b)]
    assert vo.prefetch(a, is_train=False)
    got_a = vo(a, is_train=False)['motion'].tensor().clone()
    assert vo.prefetch(b, is_train=False)
    _ = torch.randn(512, 512, device=cuda) @ torch.randn(512, 512, device=cuda)      # unrelated work on the main stream
    got_b = vo(b, is_train=False)['motion'].tensor().clone()
    tol = dict(rtol=1e-5, atol=1e-6)                # MIOpen's split-K convolutions are not bit-reproducible run to run
    torch.testing.assert_close(got_a, ref[0], **tol)
    torch.testing.assert_close(got_b, ref[1], **tol)
    assert float((ref[0] - ref[1]).abs().max()) > 1e-4                                # the two batches do differ
    # a sample that was not prefetched is computed inline; a trainable flow net refuses to prefetch
    torch.testing.assert_close(vo(a, is_train=False)['motion'].tensor(), ref[0], **tol)
    for p in vo.vonet.flowNet.parameters():
        p.requires_grad_(True)
    assert vo.prefetch(a) is False


def test_graph_replay_of_the_frozen_forward(cuda):
    """VONet.set_graph_frozen: the frozen flow + disparity forward replayed from a captured HIP graph gives what the
    launch-by-launch forward gives -- for new inputs, from a side stream (prefetch), and train-mode BatchNorm statistics
    keep being updated by the replays."""
    from islam_amd.TartanVO import TartanVO
    torch.manual_seed(0)
    vo = TartanVO(correct_scale=False, fix_parts=('flow', 'stereo'), use_kitti_coord=True, frozen_dtype=torch.bfloat16,
                  flow_dtype=torch.bfloat16)
    with torch.no_grad():
        vo.vonet.stereoNet.conv_c13.weight.zero_()
        vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
    a, b = synthetic.stereo_batch(2, seed=3), synthetic.stereo_batch(2, seed=4)
    imgs = lambda s: [s[k].to(cuda) for k in ('img0', 'img1', 'img0_norm', 'img0_r_norm')]
    vo.vonet.eval()
    with torch.no_grad():
        ref = [tuple(t.clone() for t in vo.vonet.frozen_forward(*imgs(s))) for s in (a, b)]
    vo.vonet.set_graph_frozen(True)
    with torch.no_grad():
        got_a = vo.vonet.frozen_forward(*imgs(a))           # captures
        got_b = vo.vonet.frozen_forward(*imgs(b))           # replays with other inputs
        got_a2 = vo.vonet.frozen_forward(*imgs(a))
    assert len(vo.vonet._graphs) == 1
    tol = dict(rtol=2e-2, atol=2e-2)                        # bf16 nets; MIOpen's kernels are not bit-reproducible run to run
    for g, r in ((got_a, ref[0]), (got_b, ref[1]), (got_a2, ref[0])):
        torch.testing.assert_close(g[0], r[0], **tol)
        torch.testing.assert_close(g[1], r[1], **tol)
    assert float((ref[0][0] - ref[1][0]).abs().max()) > 0.05         # the two batches do differ
    assert got_a[0].data_ptr() != got_a2[0].data_ptr()               # results are copies, not the graph's static buffers
    # through prefetch on the side stream, and the whole forward
    ref_motion = vo(a, is_train=False)['motion'].tensor().clone()
    assert vo.prefetch(a, is_train=False)
    torch.testing.assert_close(vo(a, is_train=False)['motion'].tensor(), ref_motion, rtol=1e-3, atol=1e-4)
    # train mode is a different graph; its replays update the running statistics of the master's BatchNorm layers
    bn = vo.vonet.stereoNet.feature_extraction.firstconv[0][1]
    vo.vonet.train()
    with torch.no_grad():
        vo.vonet.frozen_forward(*imgs(a))
        before = bn.running_mean.clone()
        vo.vonet.frozen_forward(*imgs(b))
    assert len(vo.vonet._graphs) == 2
    assert float((bn.running_mean - before).abs().max()) > 0
    vo.vonet.reset_graphs()
    assert len(vo.vonet._graphs) == 0
    # destroy the captured graphs and their private memory pool HERE, with the device idle, not whenever the garbage
    # collector gets to them in the middle of a later test's launches
    torch.cuda.synchronize()
    del vo, got_a, got_b, got_a2, ref
    import gc
    gc.collect()
    torch.cuda.synchronize()


def _loop_closure_problem(F, closures, seed=5):
    """chain_problem(F) with some chain edges replaced by long-range ones (loop closures), measured with a little noise."""
    prob, tr = chain_problem(F)
    links = prob['links'].copy()
    vo = prob['vo_motions'].copy()
    gt = np.concatenate([tr['gt_pos'], tr['gt_quat']], 1)
    rng = np.random.default_rng(seed)
    for e, (i, j) in closures.items():
        links[e] = (i, j)
        rel = lie.se3_mul(lie.se3_inv(gt[i]), gt[j])
        vo[e] = lie.se3_mul(rel, lie.se3_exp(rng.normal(0, 0.01, 6)))
    return dict(prob, links=links, vo_motions=vo)


def test_general_topology_band_pcg_equals_dense(cuda):
    """run_lm_band_pcg (block-tridiagonal solver + conjugate gradients on the off-band blocks of the loop closures) solves
    the SAME normal equations as the dense path: identical LM trajectory, and both match the oracle's dense LM."""
    from islam_amd import lietensor as pp
    from islam_amd.pvgo import run_pvgo
    p2 = _loop_closure_problem(33, {3: (0, 9), 11: (4, 17), 19: (30, 2), 25: (25, 26), 28: (31, 8)})
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    out = {}
    for how in ('dense', 'band_pcg'):
        out[how] = run_pvgo(pp.SE3(t(p2['init_nodes'])), t(p2['init_vels']), pp.SE3(t(p2['vo_motions']).to(cuda)),
                            torch.tensor(p2['links']), t(p2['dts']), pp.SO3(t(p2['imu_drots'])), t(p2['imu_dtrans']),
                            t(p2['imu_dvels']), device='cuda', loss_weight=LW, general_solver=how, return_info=True)
    info_d, info_p = out['dense'][5], out['band_pcg'][5]
    assert info_p['off_band_edges'] == 4 and 0 < info_p['pcg_iterations'] <= info_p['trials'] * (12 * 4 + 11)
    assert info_p['steps'] == info_d['steps'] and info_p['trials'] == info_d['trials']
    np.testing.assert_allclose([x[0] for x in info_p['trace']], [x[0] for x in info_d['trace']], rtol=1e-9)
    np.testing.assert_allclose(out['band_pcg'][2].tensor().numpy(), out['dense'][2].tensor().numpy(), atol=1e-9)
    np.testing.assert_allclose(out['band_pcg'][3].numpy(), out['dense'][3].numpy(), atol=1e-9)
    otl, orl, on, ov, _ = opvgo.run_pvgo(**p2, loss_weight=LW, mode='dense')
    err = se3_log_err(out['band_pcg'][2].tensor().numpy(), on)
    assert (err / np.maximum(np.linalg.norm(lie.se3_log(on), axis=-1), 1e-6)).max() < 1e-6


def test_general_topology_long_chain_with_loop_closures(cuda):
    """A 3000-node trajectory with six loop closures: the automatic choice is the band + PCG path (no (9N)^2 matrix); it
    agrees with the dense path on the same problem."""
    from islam_amd import lietensor as pp
    from islam_amd.pvgo import run_pvgo
    F = 3000
    closures = {100: (0, 2900), 700: (350, 2100), 1300: (1299, 40), 1900: (2500, 600), 2400: (10, 1500), 2800: (2999, 1)}
    p2 = _loop_closure_problem(F, closures, seed=9)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    args = (pp.SE3(t(p2['init_nodes'])), t(p2['init_vels']), pp.SE3(t(p2['vo_motions']).to(cuda)), torch.tensor(p2['links']),
            t(p2['dts']), pp.SO3(t(p2['imu_drots'])), t(p2['imu_dtrans']), t(p2['imu_dvels']))
    auto = run_pvgo(*args, device='cuda', loss_weight=LW, return_info=True)
    assert auto[5]['off_band_edges'] == 6 and auto[5]['pcg_iterations'] > 0          # band + PCG was chosen
    assert auto[5]["trace"][-1][0] <= auto[5]["trace"][0][0]                          # the loss went down
    dense = run_pvgo(*args, device='cuda', loss_weight=LW, return_info=True, general_solver='dense')
    assert dense[5]['trials'] == auto[5]['trials']
    np.testing.assert_allclose(auto[2].tensor().numpy(), dense[2].tensor().numpy(), atol=1e-7)
    np.testing.assert_allclose(auto[3].numpy(), dense[3].numpy(), atol=1e-7)
