"""GPU tests of the IMU path with the denoiser in the loop (reference imu_integrator.py:60-65,107-113 +
Network/IMUDenoiseNet.py:28-62) and of the IMU-target loss (pvgo.py:95-111, `target='imu'`)."""
import os

import numpy as np
import pytest
import torch

from oracle import imu as oimu, lie, pvgo as opvgo
from tests.golden.netfill import fill_state_dict, make_input
from tests.helpers import chain_problem

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
LW = (1, 0.1, 10, 0.1)


def _stream():
    """An IMU stream whose frames [0, 9) cover exactly the 83 samples of the golden denoiser fixture: eight frames of ten
    samples and a last one of three (Q14: the correction of the last conv step is stretched over the remainder), followed by
    two frames of 4 samples each (Q13: fewer than 10 samples in a batch -> the denoiser is skipped)."""
    acc, gyro = make_input('acc').numpy().astype(np.float64), make_input('gyro').numpy().astype(np.float64) * 0.05
    rng = np.random.default_rng(3)
    acc = np.concatenate([acc, rng.normal(size=(8, 3))])
    gyro = np.concatenate([gyro, rng.normal(size=(8, 3)) * 0.05])
    sync = np.array([0, 10, 20, 30, 40, 50, 60, 70, 80, 82, 86, 90])
    return acc, gyro, np.full(len(acc), 0.01), sync


@pytest.mark.parametrize('denoise_gyro', [False, True])
def test_imu_module_with_denoiser_matches_reference_vectors(cuda, tmp_path, denoise_gyro):
    """IMUModule(denoise_model_name=...) on the GPU: the denoised stream is the reference-generated golden vector
    (tests/golden/nets_denoise.npz, made with gyro = make_input('gyro')), integrated by the oracle."""
    from islam_amd import nets
    from islam_amd.imu_integrator import IMUModule
    ref = np.load(os.path.join(G, 'nets_denoise.npz'))
    ckpt = str(tmp_path / 'imudenoise.pkl')
    torch.save(fill_state_dict(nets.IMUCorrector_CNN_GRU_WO_COV()).state_dict(), ckpt)
    acc, gyro, dts, sync = _stream()
    gyro[:83] = make_input('gyro').numpy()                      # the fixture's gyro input, unscaled
    init = dict(pos=np.zeros(3), rot=np.array([0, 0, 0, 1.0]), vel=np.array([1.0, 0, 0]))
    mod = IMUModule(acc, gyro, dts, np.zeros(3), np.zeros(3), init, 9.81, sync, device='cuda', denoise_model_name=ckpt,
                    denoise_accel=True, denoise_gyro=denoise_gyro, dtype=torch.float32)
    assert mod.use_denoise_model and not mod.optm_bias
    # frames 0..9 = samples [0, 83): the denoiser runs (83 >= 10), remainder stretch included
    den_a = ref['cacc'].astype(np.float64)
    den_g = ref['cgyro'].astype(np.float64) if denoise_gyro else gyro[:83]
    for motion in (False, True):
        pos, rot, covs, vel = mod.integrate(0, 9, init, motion_mode=motion)
        want = oimu.integrate(den_a, den_g, dts[:83], sync[:10], 0, 9, init, 9.81, motion, dtype=np.float32)
        np.testing.assert_allclose(pos.numpy(), want[0], rtol=2e-4, atol=2e-5)       # MIOpen GRU vs the CPU GRU: ~1e-6 in
        np.testing.assert_allclose(rot.tensor().numpy(), want[1], rtol=2e-4, atol=2e-5)   # the corrected samples
        np.testing.assert_allclose(vel.numpy(), want[2], rtol=2e-4, atol=2e-5)
        raw = oimu.integrate(acc[:83], gyro[:83], dts[:83], sync[:10], 0, 9, init, 9.81, motion, dtype=np.float32)
        assert np.abs(vel.numpy() - raw[2]).max() > 1e-3          # ... and the correction is not a no-op
    # frames 9..11 = 8 samples: imu_integrator.py:107 skips the denoiser -> raw stream, bit-exact
    for motion in (False, True):
        pos, rot, covs, vel = mod.integrate(9, 11, init, motion_mode=motion)
        want = oimu.integrate(acc, gyro, dts, sync, 9, 11, init, 9.81, motion, dtype=np.float32)
        np.testing.assert_array_equal(pos.numpy(), want[0])
        np.testing.assert_array_equal(rot.tensor().numpy(), want[1])
        np.testing.assert_array_equal(vel.numpy(), want[2])


def test_run_pvgo_imu_target_matches_oracle(cuda):
    """run_pvgo(target='imu') (train.py:262 in IMU epochs): the per-frame IMU losses of pvgo.py:95-111."""
    from islam_amd import lietensor as pp
    from islam_amd.pvgo import run_pvgo
    prob, _ = chain_problem(9)
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
    tl, rl, nodes, vels, covs = run_pvgo(pp.SE3(f32(prob['init_nodes'])), f32(prob['init_vels']), pp.SE3(f32(prob['vo_motions']).to(cuda)),
                                         torch.tensor(prob['links']), f32(prob['dts']), pp.SO3(f32(prob['imu_drots'])),
                                         f32(prob['imu_dtrans']), f32(prob['imu_dvels']), device='cuda', radius=1e4,
                                         loss_weight=LW, target='imu')
    p32 = {k: (np.asarray(v, np.float32).astype(np.float64) if k != 'links' else v) for k, v in prob.items()}
    otl, orl, on, ov, _ = opvgo.run_pvgo(**p32, loss_weight=LW, mode='dense', target='imu')
    assert tl.shape == (8,) and rl.shape == (8,) and tl.device.type == 'cuda'
    # float32 evaluation of losses ~1e-6..1e-3: differences of O(1) velocities squared
    np.testing.assert_allclose(tl.detach().cpu().numpy(), otl, rtol=5e-3, atol=2e-6)
    np.testing.assert_allclose(rl.detach().cpu().numpy(), orl, rtol=5e-3, atol=1e-9)
    d = lie.se3_log(lie.se3_mul(lie.se3_inv(on), nodes.numpy().astype(np.float64)))
    assert (np.linalg.norm(d, axis=1) / np.maximum(np.linalg.norm(lie.se3_log(on), axis=1), 1e-6)).max() < 1e-4
