"""The torch-op restatement of pp.module.IMUPreintegrator (tests/golden/make_imu_torchops_golden.py; reference
imu_integrator.py:55-56,146) against the C oracle that defines the HIP kernel's floating-point contract: the fixtures are
reproducible from the committed script, and the two arithmetics agree to a few ulps (the gap tests/test_imu_gpu.py bounds for the
kernel itself)."""
import os

import numpy as np
import pytest

from oracle import cwrap

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('tag,dtype', [('f64', np.float64), ('f32', np.float32)])
def test_c_oracle_is_within_ulps_of_the_torch_op_sequence(tag, dtype):
    z = np.load(os.path.join(GOLD, 'imu_torchops_%s.npz' % tag))
    eps, worst, equal, total = np.finfo(dtype).eps, 0.0, 0, 0
    for name in ('ragged', 'car9_10', 'car33_7'):
        for motion in (False, True):
            ref = cwrap.imu_integrate(z[name + '_dts'], z[name + '_gyros'], z[name + '_accels'], z[name + '_seg'], z[name + '_init_pos'],
                                      z[name + '_init_rot'], z[name + '_init_vel'], float(z[name + '_gravity']), motion, dtype)
            m = name + ('_motion' if motion else '_world')
            for r, k in zip(ref, ('pos', 'rot', 'vel')):
                g = z[m + '_' + k]
                scale = np.ones((len(g), 1)) if k == 'rot' else np.maximum(np.abs(g).max(axis=-1, keepdims=True), np.finfo(dtype).tiny)
                worst = max(worst, float((np.abs(r.astype(np.float64) - g.astype(np.float64)) / (eps * scale)).max()))
                equal += int((r == g).sum())
                total += g.size
    assert worst <= 4.0, worst
    assert equal / total > 0.6            # most entries are bit-equal; the rest differ in the last place


def test_fixture_regenerates_bit_identically():
    import torch
    from tests.golden import make_imu_torchops_golden as mk
    z = np.load(os.path.join(GOLD, 'imu_torchops_f64.npz'))
    if str(z['torch_version']) != torch.__version__:
        pytest.skip('fixture written by torch %s' % z['torch_version'])
    for name, tr, seg in mk.cases():
        init = tr['init'] if 'init' in tr else dict(pos=tr['gt_pos'][0], rot=tr['gt_quat'][0], vel=tr['gt_vel'][0])
        S = int(seg[-1])
        pos, rot, vel = mk.integrate(tr['imu_dts'][:S], tr['gyros'][:S], tr['accels'][:S], seg, init, float(tr['gravity']), True, torch.float64)
        np.testing.assert_array_equal(pos, z[name + '_motion_pos'])
        np.testing.assert_array_equal(rot, z[name + '_motion_rot'])
