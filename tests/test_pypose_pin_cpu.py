"""Pins oracle/ and the host-side LieTensor shim to the reference run under REAL PyPose (fixtures from
tests/golden/make_pvgo_golden.py).  XFAIL "parity unpinned" while the fixtures do not exist."""
import numpy as np
import pytest
import torch

from oracle import imu as oimu, lie, pvgo as opvgo
from tests import pin


def _rel_se3(X, Xref):
    d = lie.se3_log(lie.se3_mul(lie.se3_inv(Xref), X))
    return (np.linalg.norm(d, axis=-1) / np.maximum(np.linalg.norm(lie.se3_log(Xref), axis=-1), 1e-6)).max()


@pytest.mark.parametrize('case', pin.PVGO_CASES)
def test_oracle_lm_matches_pypose(case):
    """reference pvgo.py:122-205 under pp.optim.LM vs oracle/pvgo.py: poses to 1e-4 rel on the SE(3) log (north_star), the loss
    the scheduler sees after every step, the final per-edge losses and their gradient."""
    fx = pin.fixture('pvgo_%s.npz' % case)
    prob = pin.pvgo_inputs(fx)
    p32 = {k: (np.asarray(v, np.float32).astype(np.float64) if k != 'links' else v) for k, v in prob.items()}
    lw = tuple(fx['loss_weight'])
    tl, rl, nodes, vels, _, opt = opvgo.run_pvgo(**p32, loss_weight=lw, mode='dense', return_optimizer=True)
    assert _rel_se3(nodes, fx['nodes'].astype(np.float64)) < 1e-4
    np.testing.assert_allclose(vels, fx['vels'], rtol=1e-4, atol=1e-5)
    # the loss optimizer.step() returned at every scheduler step: same NUMBER of steps (StopOnPlateau) and same values
    assert len(opt.step_losses) == len(fx['step_losses'])
    np.testing.assert_allclose(opt.step_losses, fx['step_losses'], rtol=1e-3)
    np.testing.assert_allclose(tl, fx['trans_loss'], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(rl, fx['rot_loss'], rtol=2e-3, atol=1e-9)
    og = opvgo.vo_loss_grad(opt.nodes, prob['links'], p32['vo_motions'], np.full(len(tl), 0.1), np.full(len(tl), 1.0))
    np.testing.assert_allclose(og, fx['vo_grad'], rtol=5e-3, atol=1e-5)


def test_oracle_imu_target_matches_pypose():
    fx = pin.fixture('pvgo_chain9_imu.npz')
    prob = pin.pvgo_inputs(fx)
    p32 = {k: (np.asarray(v, np.float32).astype(np.float64) if k != 'links' else v) for k, v in prob.items()}
    tl, rl, nodes, vels, _ = opvgo.run_pvgo(**p32, loss_weight=tuple(fx['loss_weight']), mode='dense', target='imu')
    np.testing.assert_allclose(tl, fx['trans_loss'], rtol=5e-3, atol=2e-6)
    np.testing.assert_allclose(rl, fx['rot_loss'], rtol=5e-3, atol=1e-9)


@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_oracle_imu_preintegration_matches_pypose(tag):
    """imu_integrator.py:69-164 under pp.module.IMUPreintegrator vs oracle/imu_preint.c: bit-equal in float64 (north_star),
    frame intervals with 0 / 1 / 10 / 70 samples, both modes."""
    fx = pin.fixture('imu_%s.npz' % tag)
    dt = np.float64 if tag == 'f64' else np.float32
    init = dict(pos=fx['init_pos'], rot=fx['init_rot'], vel=fx['init_vel'])
    for motion, m in ((False, 'world'), (True, 'motion')):
        pos, rot, vel = oimu.integrate(fx['accels'], fx['gyros'], fx['dts'], fx['sync'], 0, len(fx['sync']) - 1, init,
                                       float(fx['gravity']), motion, dtype=dt)
        if tag == 'f64':
            np.testing.assert_array_equal(pos, fx[m + '_pos'])
            np.testing.assert_array_equal(rot, fx[m + '_rot'])
            np.testing.assert_array_equal(vel, fx[m + '_vel'])
        else:
            np.testing.assert_allclose(pos, fx[m + '_pos'], rtol=2e-6, atol=2e-6)
            np.testing.assert_allclose(rot, fx[m + '_rot'], rtol=2e-6, atol=2e-7)
            np.testing.assert_allclose(vel, fx[m + '_vel'], rtol=2e-6, atol=2e-6)


def test_lie_ops_and_autograd_conventions_match_pypose():
    """SURVEY Appendix C items 1-10 on the oracle (values) and on islam_amd/lietensor.py (values + gradients)."""
    from islam_amd import lietensor as pp, transformation as tf
    fx = pin.fixture('lieops.npz')
    X, Y, pts = fx['X'], fx['Y'], fx['pts']
    np.testing.assert_allclose(lie.se3_exp(fx['xi']), X, atol=1e-13)
    np.testing.assert_allclose(lie.se3_mul(X, Y), fx['val_mul'], atol=1e-13)
    np.testing.assert_allclose(lie.se3_inv(X), fx['val_inv'], atol=1e-13)
    np.testing.assert_allclose(lie.se3_log(X), fx['val_log'], atol=1e-12)
    np.testing.assert_allclose(lie.se3_act(X, pts), fx['val_act'], atol=1e-13)
    np.testing.assert_allclose(lie.so3_log(X[:, 3:]), fx['val_rot_log'], atol=1e-13)
    d = fx['add_delta']
    np.testing.assert_allclose(lie.se3_mul(lie.se3_exp(d[:, :6]), X), fx['val_add'], atol=1e-13)      # LieTensor.add_
    # Hillis-Steele cumprod, float32: association order visible in the last bits
    q = fx['cumprod_in'].astype(np.float32)
    n, s = len(q), 1
    while s < n:
        prev = q.copy()
        q[s:] = lie.quat_mul(prev[:-s], prev[s:]).astype(np.float32)
        s *= 2
    np.testing.assert_allclose(q, fx['cumprod_out'], rtol=0, atol=2e-7)
    # the shim: values and PyPose's gradient convention
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    w6, w3 = t(fx['w6']), t(fx['w3'])

    def grad_of(fn, *leaves):
        ls = [t(l).requires_grad_(True) for l in leaves]
        fn(*ls).backward()
        return [l.grad.numpy() for l in ls]
    np.testing.assert_allclose(grad_of(lambda A: (pp.SE3(A).Log().tensor() * w6).sum(), X)[0], fx['g_log'], atol=1e-10)
    gl, gr = grad_of(lambda A, B: ((pp.SE3(A) @ pp.SE3(B)).Log().tensor() * w6).sum(), X, Y)
    np.testing.assert_allclose(gl, fx['g_mul_left'], atol=1e-10)
    np.testing.assert_allclose(gr, fx['g_mul_right'], atol=1e-10)
    np.testing.assert_allclose(grad_of(lambda A: (pp.SE3(A).Inv().Log().tensor() * w6).sum(), X)[0], fx['g_inv'], atol=1e-10)
    np.testing.assert_allclose(grad_of(lambda A: ((pp.SE3(A) @ t(pts)) * w3).sum(), X)[0], fx['g_act'], atol=1e-10)
    np.testing.assert_allclose(grad_of(lambda v: ((pp.se3(v).Exp() @ pp.SE3(t(Y))).Log().tensor() * w6).sum(), fx['xi'])[0],
                               fx['g_exp'], atol=1e-10)
    np.testing.assert_allclose(grad_of(lambda a: (pp.SO3(a).Log().tensor() * w3).sum(), fx['q'])[0], fx['g_so3_log'], atol=1e-10)
    np.testing.assert_allclose(grad_of(lambda v: (pp.so3(v).Exp().Log().tensor() * w3).sum(), fx['xi'][:, 3:])[0],
                               fx['g_so3_exp'], atol=1e-10)
    # Datasets/transformation.py helpers
    m6 = t(fx['tf_in'])
    np.testing.assert_allclose(tf.cvtSE3_pypose(m6).tensor().numpy(), fx['tf_cvt'], atol=1e-13)
    K = tf.tartan2kitti_pypose(m6)
    np.testing.assert_allclose(K.tensor().numpy(), fx['tf_kitti'], atol=1e-13)
    P = tf.motion2pose_pypose(K, pp.SE3(t(X[0])))
    np.testing.assert_allclose(P.tensor().numpy(), fx['tf_poses'], atol=1e-13)
    np.testing.assert_allclose(tf.pose2motion_pypose(P).tensor().numpy(), fx['tf_motions'], atol=1e-13)
