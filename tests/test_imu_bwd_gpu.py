"""GPU tests of the differentiable IMU path: islam_imu_preint_bwd against plain torch autograd
through a restatement of the frame loop, and the IMU-target epoch reaching the denoiser's parameters
(reference imu_integrator.py:107-113,146-153; pvgo.py:95-111; train.py:177-179,207-212)."""
import numpy as np
import pytest
import torch

from islam_amd import synthetic
from oracle import lie
from tests.golden.netfill import fill_state_dict
from tests.helpers import imu_preint_torch, tq_log

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-10), (torch.float32, 2e-4)])
@pytest.mark.parametrize('motion', [True, False])
def test_preintegration_backward_matches_autograd(cuda, dtype, tol, motion):
    from islam_amd import lietensor as pp, ops
    rng = np.random.default_rng(0)
    seg = np.array([0, 4, 4, 9, 12, 12, 15, 85])              # empty frames (Q12), short ones, one of 70 samples
    S = 85
    dt = torch.tensor(rng.uniform(0.005, 0.02, S))
    gyro = torch.tensor(rng.normal(size=(S, 3)) * 0.5)
    acc = torch.tensor(rng.normal(size=(S, 3)) * 2)
    q0, p0, v0 = torch.tensor(lie.so3_exp(np.array([0.3, -0.2, 0.5]))), torch.tensor([1.0, 2, 3]).double(), torch.tensor([0.5, -1, 0.2]).double()
    rows = len(seg) - 1 if motion else len(seg)
    wp, wv, wr = (torch.tensor(rng.normal(size=(rows, 3))) for _ in range(3))
    # reference: float64 torch autograd, raw-quaternion Log
    g_ref, a_ref = gyro.clone().requires_grad_(True), acc.clone().requires_grad_(True)
    P, R, V = imu_preint_torch(dt, g_ref, a_ref, seg, p0, q0, v0, 9.81, motion)
    ((P * wp).sum() + (V * wv).sum() + (tq_log(R) * wr).sum()).backward()
    # HIP forward + backward; the rotation rows go through the LieTensor shim (left-tangent gradients, PyPose's convention)
    d = lambda t: t.to(cuda, dtype)
    g_dev, a_dev = d(gyro).requires_grad_(True), d(acc).requires_grad_(True)
    segt = torch.tensor(seg, dtype=torch.int64)
    pos, rot, vel = ops.imu_preint(d(dt), g_dev, a_dev, segt.to(cuda), seg, d(p0), d(q0), d(v0), 9.81, motion)
    np.testing.assert_allclose(pos.detach().cpu().numpy(), P.detach().numpy(), rtol=0, atol=1e-12 if dtype == torch.float64 else 1e-5)
    loss = (pos * d(wp)).sum() + (vel * d(wv)).sum() + (pp.SO3(rot).Log().tensor() * d(wr)).sum()
    loss.backward()
    for got, ref in ((g_dev.grad, g_ref.grad), (a_dev.grad, a_ref.grad)):
        scale = float(ref.abs().max())
        assert float((got.double().cpu() - ref).abs().max()) <= tol * scale, (float((got.double().cpu() - ref).abs().max()), scale)
    # rows of empty frames and of no frame at all get exactly zero
    assert torch.count_nonzero(g_dev.grad[:S]) > 0


def test_imu_epoch_gradient_reaches_the_denoiser(cuda, tmp_path):
    """IMUModule(train_denoiser=True) + run_pvgo(target='imu'): d loss / d denoiser-parameters is non-zero and equals the
    directional finite difference of the same loss; with the reference's eval=True behaviour it is absent (F6)."""
    from islam_amd import lietensor as pp, nets
    from islam_amd.imu_integrator import IMUModule
    from islam_amd.pvgo import run_pvgo
    from tests.helpers import chain_problem
    ckpt = str(tmp_path / 'imudenoise.pkl')
    torch.manual_seed(0)
    den = fill_state_dict(nets.IMUCorrector_CNN_GRU_WO_COV())
    with torch.no_grad():                                        # small corrections, like a trained denoiser
        den.pose_decoder[2].weight.mul_(0.05)
        den.pose_decoder[2].bias.mul_(0.05)
    torch.save(den.state_dict(), ckpt)
    B = 8
    tr = synthetic.car_trajectory(B + 1, seed=9)
    prob, _ = chain_problem(B + 1, seed=9)
    mod = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], np.zeros(3), np.zeros(3), tr['init'], tr['gravity'], tr['rgb2imu_sync'],
                    device='cuda', denoise_model_name=ckpt, denoise_accel=True, denoise_gyro=True, dtype=torch.float64)
    mod.denoiser.double()
    f64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)

    def loss_fn():
        it, ir, _, iv = mod.integrate(0, B, tr['init'], motion_mode=False)
        dtr, drot, _, dvel = mod.integrate(0, B, tr['init'], motion_mode=True)
        nodes = pp.SE3(torch.cat((it, ir.tensor()), 1))
        tl, rl, *_ = run_pvgo(nodes, iv, pp.SE3(f64(prob['vo_motions']).to(cuda)), torch.tensor(prob['links']), f64(prob['dts']), drot, dtr, dvel,
                              device='cuda', radius=1e4, loss_weight=(1, 0.1, 10, 0.1), target='imu')
        return torch.cat((1.0 * rl, 0.1 * tl)).sum()
    # reference behaviour: no gradient path at all
    assert not loss_fn().requires_grad
    mod.train_denoiser = True
    L = loss_fn()
    assert L.requires_grad
    L.backward()
    params = [p for p in mod.denoiser.parameters()]
    grads = [p.grad.clone() for p in params]
    assert all(g is not None and torch.isfinite(g).all() for g in grads) and sum(float(g.abs().sum()) for g in grads) > 0
    # Finite-difference check.  The reference's imu_loss (pvgo.py:95-111) is evaluated on graph.nodes AFTER the optimisation
    # and differentiates them as constants (no implicit differentiation through LM): the gradient is the PARTIAL derivative
    # w.r.t. the IMU deltas at fixed nodes.  So: the optimised (un-aligned) nodes of the same LM run, held fixed, and the
    # oracle's imu_loss on the deltas integrated with the parameter nudged along the gradient direction.
    from islam_amd import ops
    from oracle import pvgo as opvgo
    with torch.no_grad():
        it, ir, _, iv = mod.integrate(0, B, tr['init'], motion_mode=False)
        dtr, drot, _, dvel = mod.integrate(0, B, tr['init'], motion_mode=True)
        nodes, vels = torch.cat((it, ir.tensor()), 1).to(cuda), iv.to(cuda)
        dv = lambda t: pp._plain(t).to(cuda, torch.float64).contiguous()
        ops.pvgo_run_chain(nodes, vels, f64(prob['vo_motions']).to(cuda), dv(drot), dv(dtr), dv(dvel), f64(prob['dts']).to(cuda),
                           ops.pvgo_default_params((1, 0.1, 10, 0.1), radius=1e4))
    nfix, vfix = nodes.cpu().numpy(), vels.cpu().numpy()

    def partial_loss():
        with torch.no_grad():
            _, drot, _, dvel = mod.integrate(0, B, tr['init'], motion_mode=True)
        tl, rl = opvgo.imu_loss(nfix, vfix, drot.tensor().numpy(), dvel.numpy())
        return float(rl.sum() + 0.1 * tl.sum())
    assert partial_loss() == pytest.approx(float(L), rel=1e-6)
    p = mod.denoiser.pose_decoder[2].weight
    gdir = p.grad / p.grad.norm()
    h = 1e-5
    with torch.no_grad():
        p.add_(h * gdir)
        Lp = partial_loss()
        p.sub_(2 * h * gdir)
        Lm = partial_loss()
        p.add_(h * gdir)
    fd, an = (Lp - Lm) / (2 * h), float((p.grad * gdir).sum())
    assert an > 0 and fd == pytest.approx(an, rel=1e-4), (fd, an)


def test_bilevel_loop_alternates_vo_and_imu_epochs_and_writes_snapshots(cuda, tmp_path):
    """train.py:163,172-198,207-212,51-61: a 'vo' epoch (VO forward, pose-head gradient, Adam step, seven snapshot files), then
    an 'imu' epoch that reuses the stored VO motions (no VO forward) and -- with the F6 fix switched on -- trains the
    denoiser."""
    from islam_amd import lietensor as pp, nets
    from islam_amd.TartanVO import TartanVO
    from islam_amd.bilevel import BilevelLoop
    from islam_amd.imu_integrator import IMUModule
    torch.manual_seed(0)
    B = 2
    ckpt = str(tmp_path / 'imudenoise.pkl')
    den = fill_state_dict(nets.IMUCorrector_CNN_GRU_WO_COV())
    with torch.no_grad():
        den.pose_decoder[2].weight.mul_(0.05)
        den.pose_decoder[2].bias.mul_(0.05)
    torch.save(den.state_dict(), ckpt)
    vo = TartanVO(correct_scale=False, fix_parts=('flow', 'stereo'), use_kitti_coord=True)
    with torch.no_grad():
        vo.vonet.stereoNet.conv_c13.weight.zero_()
        vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
    tr = synthetic.car_trajectory(2 * B + 1, seed=9)
    imu = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], np.zeros(3), np.zeros(3), tr['init'], tr['gravity'],
                    tr['rgb2imu_sync'], device='cuda', denoise_model_name=ckpt, denoise_accel=True, denoise_gyro=True)
    loop = BilevelLoop(vo, imu, pp.identity_SE3(), tr['init'], batch_size=B, train_imu_denoiser=True)
    assert loop.imu_optimizer is not None
    samples = []
    for k in range(2):
        s = synthetic.stereo_batch(B, seed=100 + k)
        s['link'] = s['link'] + k * B
        samples.append(s)
        assert np.isfinite(loop.step(s, target='vo'))
    assert all(p.grad is None for p in imu.denoiser.parameters())               # VO epochs never touch the denoiser
    root = str(tmp_path / 'train')
    loop.end_epoch('vo', trainroot=root, epoch=1, reset=True)
    shapes = {'vo_pose': (2 * B + 1, 7), 'vo_motion': (2 * B, 7), 'pgo_pose': (2 * B + 1, 7), 'pgo_motion': (2 * B, 7),
              'pgo_vel': (2 * B + 1, 3), 'imu_pose': (2 * B + 1, 7), 'imu_motion': (2 * B, 7)}
    for name, shp in shapes.items():
        assert np.loadtxt('%s/1/%s.txt' % (root, name)).shape == shp, name
    assert loop.prev_vo_motions.shape == (2 * B, 7) and loop.current_idx == 0 and len(loop.pgo_poses) == 1
    # IMU epoch: the VO network must not run
    calls = []
    orig = vo.forward
    vo.forward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    w0 = [p.detach().clone() for p in imu.denoiser.parameters()]
    pose_w0 = [p.detach().clone() for p in vo.vonet.flowPoseNet.parameters()]
    for s in samples:
        assert np.isfinite(loop.step(s, target='imu'))
    assert calls == []
    assert sum(float(p.grad.abs().sum()) for p in imu.denoiser.parameters()) > 0
    loop.end_epoch('imu', trainroot=root, epoch=2)
    assert any(not torch.equal(a, b.detach()) for a, b in zip(w0, imu.denoiser.parameters()))
    assert all(torch.equal(a, b.detach()) for a, b in zip(pose_w0, vo.vonet.flowPoseNet.parameters()))
    assert np.loadtxt('%s/2/imu_motion.txt' % root).shape == (2 * B, 7)
    # the epoch's VO motions are exactly the stored ones
    np.testing.assert_array_equal(np.loadtxt('%s/2/vo_motion.txt' % root), np.loadtxt('%s/1/vo_motion.txt' % root))
