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


def _ragged_stream(nframes, seed, p_empty=0.2):
    rng = np.random.default_rng(seed)
    counts = rng.integers(1, 6, nframes)
    counts[rng.random(nframes) < p_empty] = 0
    for i, c in ((0, 0), (7, 0), (8, 3), (255, 0), (256, 2), (257, 0), (511, 0), (512, 0)):      # block / chunk borders
        if i < nframes:
            counts[i] = c
    seg = np.concatenate([[0], np.cumsum(counts)])
    S = int(seg[-1])
    dt = rng.uniform(0.004, 0.012, S)
    gyro = rng.normal(0, 0.5, (S, 3))
    acc = rng.normal(0, 1.0, (S, 3)) + np.array([0, 0, 9.81])
    init = dict(pos=np.array([1.0, 2.0, 3.0]), rot=np.array([0.1, -0.2, 0.3, 0.9273618495495703]), vel=np.array([5.0, 0.1, -0.2]))
    return dt, gyro, acc, seg, init


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('motion', [False, True])
def test_empty_intervals_across_the_chain_blocks(cuda, dtype, motion):
    """The frame chains walk register blocks of 8 frames inside LDS chunks of 256 / 512: frames without samples (rotation and position
    held, velocity zeroed: imu_integrator.py:134-140) at block and chunk borders, a last chunk that ends inside a block."""
    dt, gyro, acc, seg, init = _ragged_stream(1301, 5)
    ref = cwrap.imu_integrate(dt, gyro, acc, seg, init['pos'], init['rot'], init['vel'], 9.81007, motion, dtype)
    out = _run(cuda, dt, gyro, acc, seg, init, 9.81007, motion, dtype)
    for o, r, name in zip(out, ref, ('pos', 'rot', 'vel')):
        np.testing.assert_array_equal(o, r, err_msg=name)


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_both_modes_from_one_pass(cuda, dtype):
    """islam_imu_preint_both == islam_imu_preint(world) + islam_imu_preint(motion), bit for bit (and the oracle)."""
    from islam_amd import ops
    dt, gyro, acc, seg, init = _ragged_stream(700, 11)
    td = {np.float64: torch.float64, np.float32: torch.float32}[dtype]
    t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=dtype), dtype=td, device=cuda)
    seg = np.ascontiguousarray(seg, dtype=np.int64)
    world, motion, _ = ops.imu_preint_both(t(dt), t(gyro), t(acc), torch.tensor(seg, device=cuda), seg, t(init['pos']), t(init['rot']),
                                           t(init['vel']), 9.81007)
    for got, mm in ((world, False), (motion, True)):
        ref = cwrap.imu_integrate(dt, gyro, acc, seg, init['pos'], init['rot'], init['vel'], 9.81007, mm, dtype)
        sep = _run(cuda, dt, gyro, acc, seg, init, 9.81007, mm, dtype)
        for g, s, r in zip(got, sep, ref):
            np.testing.assert_array_equal(g.cpu().numpy(), s)
            np.testing.assert_array_equal(s, r)


def test_imu_module_integrate_both_matches_two_calls(cuda):
    from islam_amd.imu_integrator import IMUModule
    tr = synthetic.car_trajectory(41, seed=3)
    imu = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], init=tr['init'], gravity=tr['gravity'], rgb2imu_sync=tr['rgb2imu_sync'],
                    device='cuda:0', denoise_accel=False, denoise_gyro=False, dtype=torch.float64)
    st, end = 8, 16
    init = dict(pos=tr['init']['pos'], rot=tr['init']['rot'], vel=tr['init']['vel'])
    w, m = imu.integrate_both(st, end, init)
    for got, mm in ((w, False), (m, True)):
        ref = imu.integrate(st, end, init, motion_mode=mm)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1].tensor(), ref[1].tensor()) and torch.equal(got[3], ref[3])
        assert got[2] == [] and got[0].device.type == 'cpu'
