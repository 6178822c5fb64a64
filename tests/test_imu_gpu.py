"""GPU parity of IMU pre-integration: BIT-EXACT against the plain-C oracle in float64 and float32
(floating-point contract in oracle/imu_preint.c)."""
import numpy as np
import pytest
import torch

from islam_amd import synthetic
from oracle import cwrap, imu as oimu

pytestmark = pytest.mark.gpu


def _run(cuda, dt, gyro, acc, seg, init, gravity, motion, dtype):
    from islam_amd import ops
    td = {np.float64: torch.float64, np.float32: torch.float32}[dtype]
    t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=dtype), dtype=td, device=cuda)
    seg = np.ascontiguousarray(seg, dtype=np.int64)
    zero3 = np.zeros(3)
    ip = zero3 if motion else init['pos']
    iv = zero3 if motion else init['vel']
    return [o.cpu().numpy() for o in ops.imu_preint(t(dt), t(gyro), t(acc), torch.tensor(seg, device=cuda), seg, t(ip),
                                                    t(init['rot']), t(iv), gravity, motion)]


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('motion', [False, True])
@pytest.mark.parametrize('frames,per', [(2, 10), (9, 10), (9, 1), (33, 7), (5, 70), (4, 200)])
def test_bit_exact(cuda, dtype, motion, frames, per):
    tr = synthetic.car_trajectory(frames, imu_per_frame=per, seed=frames + per)
    seg = tr['rgb2imu_sync'] - tr['rgb2imu_sync'][0]
    ref = cwrap.imu_integrate(tr['imu_dts'], tr['gyros'], tr['accels'], seg, tr['init']['pos'], tr['init']['rot'],
                              tr['init']['vel'], tr['gravity'], motion, dtype)
    out = _run(cuda, tr['imu_dts'], tr['gyros'], tr['accels'], seg, tr['init'], tr['gravity'], motion, dtype)
    for o, r, name in zip(out, ref, ('pos', 'rot', 'vel')):
        assert o.dtype == r.dtype
        np.testing.assert_array_equal(o, r, err_msg=name)


@pytest.mark.parametrize('motion', [False, True])
def test_ragged_and_empty_intervals(cuda, motion):
    """Frames with 0 IMU samples (imu_integrator.py:134-140), ragged counts, large rates (Taylor / reduction branches)."""
    rng = np.random.default_rng(0)
    counts = np.array([3, 0, 11, 1, 0, 0, 25, 2])
    seg = np.concatenate([[0], np.cumsum(counts)])
    S = int(seg[-1])
    dt = rng.uniform(0.004, 0.012, S)
    gyro = rng.normal(0, 0.5, (S, 3))
    gyro[0] = 0.0                       # theta == 0 -> Taylor branch
    gyro[5] = [400.0, -250.0, 90.0]     # |theta/2| > pi/4 -> Cody-Waite branch
    acc = rng.normal(0, 1.0, (S, 3)) + np.array([0, 0, 9.81])
    init = dict(pos=np.array([1.0, 2.0, 3.0]), rot=np.array([0.1, -0.2, 0.3, 0.9273618495495703]), vel=np.array([5.0, 0.1, -0.2]))
    for dtype in (np.float64, np.float32):
        ref = cwrap.imu_integrate(dt, gyro, acc, seg, init['pos'], init['rot'], init['vel'], 9.81007, motion, dtype)
        out = _run(cuda, dt, gyro, acc, seg, init, 9.81007, motion, dtype)
        for o, r in zip(out, ref):
            np.testing.assert_array_equal(o, r)


def test_full_size_5000_frames(cuda):
    """BASELINE config-4 size: 5000 frame intervals, 50 001 IMU samples, float64, bit-exact + chain property."""
    tr = synthetic.car_trajectory(5001)
    seg = tr['rgb2imu_sync']
    ref = oimu.integrate(tr['accels'], tr['gyros'], tr['imu_dts'], seg, 0, 5000, tr['init'], tr['gravity'], False)
    out = _run(cuda, tr['imu_dts'], tr['gyros'], tr['accels'], seg, tr['init'], tr['gravity'], False, np.float64)
    for o, r in zip(out, ref):
        np.testing.assert_array_equal(o, r)
    # world-mode rotations are the running product of motion-mode rotations (up to rounding)
    mo = _run(cuda, tr['imu_dts'], tr['gyros'], tr['accels'], seg, tr['init'], tr['gravity'], True, np.float64)
    from oracle import lie
    q = out[1][0]
    for k in range(0, 5000, 997):
        np.testing.assert_allclose(lie.quat_mul(out[1][k], mo[1][k]), out[1][k + 1], atol=1e-12)


@pytest.mark.parametrize('tag,dtype,bound', [('f64', np.float64, 4.0), ('f32', np.float32, 4.0)])
def test_gap_to_torch_op_arithmetic(cuda, tag, dtype, bound):
    """"Bit-exact" above means: to the fdlibm contract shared with oracle/imu_preint.c.  The reference runs TORCH ops
    (torch.sin / cos, cumsum, PyPose's doubling cumprod; imu_integrator.py:55-56,146): tests/golden/imu_torchops_*.npz holds that
    op sequence's results (tests/golden/make_imu_torchops_golden.py -- a restatement, not a PyPose pin).  This test REPORTS the
    maximal difference of islam_imu_preint against them in ulps of the row's largest component and bounds it."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'imu_torchops_%s.npz' % tag))
    eps = np.finfo(dtype).eps
    worst = {}
    for name in ('ragged', 'car9_10', 'car33_7'):
        init = dict(pos=z[name + '_init_pos'], rot=z[name + '_init_rot'], vel=z[name + '_init_vel'])
        for motion in (False, True):
            out = _run(cuda, z[name + '_dts'], z[name + '_gyros'], z[name + '_accels'], z[name + '_seg'], init,
                       float(z[name + '_gravity']), motion, dtype)
            m = name + ('_motion' if motion else '_world')
            for o, k in zip(out, ('pos', 'rot', 'vel')):
                g = z[m + '_' + k]
                assert o.shape == g.shape and o.dtype == g.dtype
                scale = np.ones((len(g), 1)) if k == 'rot' else np.maximum(np.abs(g).max(axis=-1, keepdims=True), np.finfo(dtype).tiny)
                ulp = float((np.abs(o.astype(np.float64) - g.astype(np.float64)) / (eps * scale)).max())
                worst[k] = max(worst.get(k, 0.0), ulp)
    print('islam_imu_preint vs torch-op arithmetic (%s): max ulp pos %.3g rot %.3g vel %.3g' % (tag, worst['pos'], worst['rot'], worst['vel']))
    assert max(worst.values()) <= bound, worst
