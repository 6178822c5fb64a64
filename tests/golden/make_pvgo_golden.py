"""Pins the PVGO / IMU / Lie-op oracle to the REAL reference by running the reference's own code under real PyPose.

  python tests/golden/make_pvgo_golden.py            # needs `import pypose` to succeed (build container only)

ONE COMMAND on a machine with network access and a checkout of sair-lab/iSLAM at /root/reference (CPU only, ~2 minutes):

  pip install "pypose>=0.6.0,<0.7" "torch>=2.0" && python tests/golden/make_pvgo_golden.py \
      && python -m pytest tests/test_pypose_pin_cpu.py -q        # then commit tests/golden/{pvgo_*,imu_*,lieops}.npz

PyPose version: the reference pins none (environment.yml lists no pypose; its README installs the current release of autumn
2024).  The API it calls -- pp.optim.LM(reject=...), pp.optim.strategy.TrustRegion, pp.optim.scheduler.StopOnPlateau,
pp.optim.solver.Cholesky, pp.module.IMUPreintegrator(prop_cov=False), pp.reprojerr, LieTensor.Jinvp / add_ / cumprod -- is the
0.6 line (0.6.0 ... 0.6.8), which is also the line oracle/pvgo.py and oracle/imu.py restate; 0.7 is excluded because it has not
been read against the oracle.  Every fixture records pypose.__version__ and torch.__version__ (keys `pypose_version`,
`torch_version`), and the consumers print them, so a pin always says what it was taken on.

Writes tests/golden/pvgo_*.npz, imu_*.npz, lieops.npz.  The consumers (tests/test_pypose_pin_cpu.py for the oracle,
tests/test_pypose_pin_gpu.py for the HIP path) use them when present and XFAIL with reason "parity unpinned" when absent.

STATUS in this build container (recorded in DESIGN.md): `pip install pypose` / `pip download pypose` find no distribution
(no network, not in /opt/wheelhouse), so this script has not run yet and PVGO / IMU parity stays "unpinned".  It is the
first thing to run on any machine that has PyPose.

What it executes (nothing of it is restated here -- these are calls INTO the reference and PyPose):
  * /root/reference/pvgo.py:122-205          run_pvgo(..., device='cpu')  (pp.optim.LM + Cholesky + TrustRegion +
                                              StopOnPlateau on PoseVelGraph), with the per-step losses the scheduler sees
                                              recorded through a wrapper around StopOnPlateau.step
  * /root/reference/imu_integrator.py:30-164  IMUModule(..., device='cpu').integrate(st, end, init, motion_mode) for frame
                                              intervals of 0 / 1 / 10 / 70 samples (pp.module.IMUPreintegrator)
  * /root/reference/Datasets/transformation.py:72-124   cvtSE3_pypose, tartan2kitti_pypose, motion2pose_pypose,
                                              pose2motion_pypose
  * PyPose's autograd conventions: gradients of Log, Mul, Inv, Exp, Act w.r.t. LieTensor
    inputs, as raw 7-/4-vectors.
Inputs are the seeded problems of tests/helpers.py (chain_problem) and tests/test_pvgo_gpu.py (_noisy_problem), stored
in the fixture next to the outputs so the consumers need not regenerate them bit-identically."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

LW = (1, 0.1, 10, 0.1)                       # run_kitti.sh:5


def _stub(name, **attrs):
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m


def noisy_problem(F, seed, sig):
    """tests/test_pvgo_gpu.py::_noisy_problem (reject-heavy LM runs)."""
    from oracle import lie
    from tests.helpers import chain_problem
    prob, _ = chain_problem(F)
    rng = np.random.default_rng(seed)
    n = prob['init_nodes'].copy()
    n[:, :3] += rng.normal(0, sig, (F, 3))
    n = lie.se3_mul(lie.se3_exp(np.concatenate([np.zeros((F, 3)), rng.normal(0, sig * 0.2, (F, 3))], 1)), n)
    return dict(prob, init_nodes=n)


def run_reference_pvgo(pp, ref_pvgo, prob, loss_weight, target='vo'):
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
    losses = []
    from pypose.optim.scheduler import StopOnPlateau
    orig = StopOnPlateau.step

    def recording_step(self, loss):
        losses.append(float(loss))
        return orig(self, loss)
    StopOnPlateau.step = recording_step
    try:
        vo = pp.SE3(f32(prob['vo_motions'])).requires_grad_(True)
        tl, rl, nodes, vels, covs = ref_pvgo.run_pvgo(
            pp.SE3(f32(prob['init_nodes'])), f32(prob['init_vels']), vo, torch.tensor(prob['links']), f32(prob['dts']),
            pp.SO3(f32(prob['imu_drots'])), f32(prob['imu_dtrans']), f32(prob['imu_dvels']), device='cpu', radius=1e4,
            loss_weight=loss_weight, target=target)
    finally:
        StopOnPlateau.step = orig
    out = dict(trans_loss=tl.detach().numpy(), rot_loss=rl.detach().numpy(), nodes=nodes.tensor().numpy(), vels=vels.numpy(),
               step_losses=np.array(losses), loss_weight=np.array(loss_weight, dtype=np.float64), target=np.array(target))
    if target == 'vo':                       # one-step back-propagation of train.py:280-283
        loss_bp = torch.cat((1.0 * rl, 0.1 * tl))
        loss_bp.backward(torch.ones_like(loss_bp))
        out['vo_grad'] = vo.grad.numpy() if vo.grad is not None else np.zeros((0,))
    for k, v in prob.items():
        out['in_' + k] = np.asarray(v)
    return out


def main():
    try:
        import pypose as pp
    except ImportError as e:
        print('pypose is not importable here (%s): PVGO / IMU parity stays UNPINNED; no fixture written.' % e)
        return 2
    _stub('cv2')
    _stub('cupy', memoize=lambda **kw: (lambda f: f))
    torch.manual_seed(0)
    import imu_integrator as ref_imu
    import pvgo as ref_pvgo
    from Datasets import transformation as ref_tf
    from islam_amd import synthetic
    from tests.helpers import chain_problem
    meta = dict(pypose_version=np.array(getattr(pp, '__version__', 'unknown')), torch_version=np.array(torch.__version__))

    # ---- 1. PVGO: chains of 9 and 65 nodes, the reject-heavy seeds, the IMU target
    cases = {'chain9': chain_problem(9)[0], 'chain65': chain_problem(65)[0],
             'noisy33': noisy_problem(33, 1, 1.5), 'noisy65a': noisy_problem(65, 8, 1.5), 'noisy65b': noisy_problem(65, 2, 1.0)}
    for name, prob in cases.items():
        np.savez_compressed(os.path.join(HERE, 'pvgo_%s.npz' % name), **run_reference_pvgo(pp, ref_pvgo, prob, LW), **meta)
    np.savez_compressed(os.path.join(HERE, 'pvgo_chain9_imu.npz'), **run_reference_pvgo(pp, ref_pvgo, cases['chain9'], LW, 'imu'), **meta)
    np.savez_compressed(os.path.join(HERE, 'pvgo_chain9_euroc.npz'),
                        **run_reference_pvgo(pp, ref_pvgo, cases['chain9'], (4, 0.1, 2, 0.1)), **meta)       # run_euroc.sh:5

    # ---- 2. IMU pre-integration: frame intervals with 0 / 1 / 10 / 70 samples, both modes, fp32 and fp64
    tr = synthetic.car_trajectory(13, seed=5)
    S = len(tr['accels'])
    sync = np.array([0, 0, 1, 11, 81, 91, 101, 101, 111], dtype=np.int64)       # 0, 1, 10, 70, 10, 10, 0, 10 samples
    assert sync[-1] < S
    init = dict(pos=tr['gt_pos'][0], rot=tr['gt_quat'][0], vel=tr['gt_vel'][0])
    for dtype, tag in ((torch.float32, 'f32'), (torch.float64, 'f64')):
        torch.set_default_dtype(dtype)
        mod = ref_imu.IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], torch.zeros(3), torch.zeros(3), init, tr['gravity'],
                                list(sync), device='cpu', denoise_model_name=None, denoise_accel=False, denoise_gyro=False)
        out = dict(accels=tr['accels'], gyros=tr['gyros'], dts=tr['imu_dts'], sync=sync, gravity=np.array(tr['gravity']),
                   init_pos=init['pos'], init_rot=init['rot'], init_vel=init['vel'])
        for motion in (False, True):
            pos, rot, covs, vel = mod.integrate(0, len(sync) - 1, init, motion_mode=motion)
            m = 'motion' if motion else 'world'
            out.update({m + '_pos': pos.numpy(), m + '_rot': rot.tensor().numpy(), m + '_vel': vel.numpy()})
        np.savez_compressed(os.path.join(HERE, 'imu_%s.npz' % tag), **out, **meta)
    torch.set_default_dtype(torch.float32)

    # ---- 3. Lie-op values and autograd conventions, float64
    g = torch.Generator().manual_seed(3)
    xi, yi = torch.randn(6, 6, generator=g, dtype=torch.float64) * 0.5, torch.randn(6, 6, generator=g, dtype=torch.float64) * 0.5
    pts = torch.randn(6, 3, generator=g, dtype=torch.float64)
    w6 = torch.linspace(0.3, 1.1, 6, dtype=torch.float64)
    w3 = torch.tensor([0.7, -0.4, 1.3], dtype=torch.float64)
    X0, Y0 = pp.se3(xi).Exp().tensor(), pp.se3(yi).Exp().tensor()
    out = dict(xi=xi.numpy(), yi=yi.numpy(), pts=pts.numpy(), X=X0.numpy(), Y=Y0.numpy(), w6=w6.numpy(), w3=w3.numpy())

    def grad_of(fn, *leaves):
        ls = [l.clone().requires_grad_(True) for l in leaves]
        fn(*ls).backward()
        return [l.grad.numpy() for l in ls]
    out['val_mul'] = (pp.SE3(X0) @ pp.SE3(Y0)).tensor().numpy()
    out['val_inv'] = pp.SE3(X0).Inv().tensor().numpy()
    out['val_log'] = pp.SE3(X0).Log().tensor().numpy()
    out['val_act'] = (pp.SE3(X0) @ pts).numpy()
    out['val_rot_log'] = pp.SE3(X0).rotation().Log().tensor().numpy()
    out['g_log'] = grad_of(lambda X: (pp.SE3(X).Log().tensor() * w6).sum(), X0)[0]
    gm = grad_of(lambda X, Y: ((pp.SE3(X) @ pp.SE3(Y)).Log().tensor() * w6).sum(), X0, Y0)
    out['g_mul_left'], out['g_mul_right'] = gm
    out['g_inv'] = grad_of(lambda X: (pp.SE3(X).Inv().Log().tensor() * w6).sum(), X0)[0]
    out['g_act'] = grad_of(lambda X: ((pp.SE3(X) @ pts) * w3).sum(), X0)[0]
    out['g_exp'] = grad_of(lambda v: ((pp.se3(v).Exp() @ pp.SE3(Y0)).Log().tensor() * w6).sum(), xi)[0]
    q0 = pp.so3(xi[:, 3:]).Exp().tensor()
    out['q'] = q0.numpy()
    out['g_so3_log'] = grad_of(lambda q: (pp.SO3(q).Log().tensor() * w3).sum(), q0)[0]
    out['g_so3_exp'] = grad_of(lambda v: (pp.so3(v).Exp().Log().tensor() * w3).sum(), xi[:, 3:].contiguous())[0]
    # LieTensor.add_ (what LM applies the step with): X <- Exp(delta[..., :6]) * X
    P = pp.Parameter(pp.SE3(X0.clone()))
    d7 = torch.cat([yi * 0.1, torch.ones(6, 1, dtype=torch.float64)], 1)
    with torch.no_grad():
        P.add_(d7)
    out['add_delta'], out['val_add'] = d7.numpy(), P.detach().tensor().numpy()
    # pp.cumprod (Hillis-Steele order, Appendix C item 10), float32 to expose the association order
    q32 = pp.so3((xi[:, 3:] * 0.3).float().repeat(3, 1)).Exp()
    out['cumprod_in'], out['cumprod_out'] = q32.tensor().numpy(), pp.cumprod(q32, dim=0, left=False).tensor().numpy()
    # transformation helpers
    m6 = torch.randn(5, 6, generator=g, dtype=torch.float64) * 0.2
    out['tf_in'] = m6.numpy()
    out['tf_cvt'] = ref_tf.cvtSE3_pypose(m6).tensor().numpy()
    K = ref_tf.tartan2kitti_pypose(m6)
    out['tf_kitti'] = K.tensor().numpy()
    Pz = ref_tf.motion2pose_pypose(K, pp.SE3(X0[0]))
    out['tf_poses'] = Pz.tensor().numpy()
    out['tf_motions'] = ref_tf.pose2motion_pypose(Pz).tensor().numpy()
    np.savez_compressed(os.path.join(HERE, 'lieops.npz'), **out, **meta)
    print('fixtures written under', HERE)
    return 0


if __name__ == '__main__':
    sys.exit(main())
