"""The HIP path against the reference run under REAL PyPose (fixtures from tests/golden/make_pvgo_golden.py): run_pvgo and
IMUModule through the drop-in surface.  XFAIL "parity unpinned" while the fixtures do not exist."""
import numpy as np
import pytest
import torch

from oracle import lie
from tests import pin

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', pin.PVGO_CASES + ('chain9_imu',))
def test_run_pvgo_matches_pypose(cuda, case):
    from islam_amd import lietensor as pp
    from islam_amd.pvgo import run_pvgo
    fx = pin.fixture('pvgo_%s.npz' % case)
    prob = pin.pvgo_inputs(fx)
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
    vo = f32(prob['vo_motions']).to(cuda).requires_grad_(True)
    target = str(fx['target'])
    tl, rl, nodes, vels, _ = run_pvgo(pp.SE3(f32(prob['init_nodes'])), f32(prob['init_vels']), pp.SE3(vo), torch.tensor(prob['links']),
                                      f32(prob['dts']), pp.SO3(f32(prob['imu_drots'])), f32(prob['imu_dtrans']), f32(prob['imu_dvels']),
                                      device='cuda', radius=1e4, loss_weight=tuple(fx['loss_weight']), target=target)
    ref = fx['nodes'].astype(np.float64)
    d = lie.se3_log(lie.se3_mul(lie.se3_inv(ref), nodes.numpy().astype(np.float64)))
    rel = np.linalg.norm(d, axis=1) / np.maximum(np.linalg.norm(lie.se3_log(ref), axis=1), 1e-6)
    assert rel.max() < 1e-4                                              # north_star: 1e-4 rel on the SE(3) log
    np.testing.assert_allclose(vels.numpy(), fx['vels'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(tl.detach().cpu().numpy(), fx['trans_loss'], rtol=5e-3, atol=2e-6)
    np.testing.assert_allclose(rl.detach().cpu().numpy(), fx['rot_loss'], rtol=5e-3, atol=1e-9)
    if target == 'vo':
        loss_bp = torch.cat((1.0 * rl, 0.1 * tl))
        loss_bp.backward(torch.ones_like(loss_bp))
        np.testing.assert_allclose(vo.grad.cpu().numpy(), fx['vo_grad'], rtol=5e-3, atol=1e-5)


@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_imu_module_matches_pypose(cuda, tag):
    from islam_amd.imu_integrator import IMUModule
    fx = pin.fixture('imu_%s.npz' % tag)
    dtype = torch.float64 if tag == 'f64' else torch.float32
    init = dict(pos=fx['init_pos'], rot=fx['init_rot'], vel=fx['init_vel'])
    mod = IMUModule(fx['accels'], fx['gyros'], fx['dts'], np.zeros(3), np.zeros(3), init, float(fx['gravity']), fx['sync'],
                    device='cuda', denoise_model_name=None, denoise_accel=False, denoise_gyro=False, dtype=dtype)
    for motion, m in ((False, 'world'), (True, 'motion')):
        pos, rot, covs, vel = mod.integrate(0, len(fx['sync']) - 1, init, motion_mode=motion)
        if tag == 'f64':                                                 # north_star: bit-matching in float64
            np.testing.assert_array_equal(pos.numpy(), fx[m + '_pos'])
            np.testing.assert_array_equal(rot.tensor().numpy(), fx[m + '_rot'])
            np.testing.assert_array_equal(vel.numpy(), fx[m + '_vel'])
        else:
            np.testing.assert_allclose(pos.numpy(), fx[m + '_pos'], rtol=2e-6, atol=2e-6)
            np.testing.assert_allclose(rot.tensor().numpy(), fx[m + '_rot'], rtol=2e-6, atol=2e-7)
            np.testing.assert_allclose(vel.numpy(), fx[m + '_vel'], rtol=2e-6, atol=2e-6)
