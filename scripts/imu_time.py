"""islam_imu_preint on the config-4 trajectory (5000 frame intervals, 50 001 samples, float64): us per call, world and motion mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from islam_amd import ops, synthetic
dev = torch.device('cuda:0')
tr = synthetic.car_trajectory(5001)
t64 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
seg_h = np.ascontiguousarray(tr['rgb2imu_sync'] - tr['rgb2imu_sync'][0], dtype=np.int64)
seg_d = torch.tensor(seg_h, device=dev)
dt, gyro, acc = t64(tr['imu_dts']), t64(tr['gyros']), t64(tr['accels'])
ip, ir, iv = t64(tr['init']['pos']), t64(tr['init']['rot']), t64(tr['init']['vel'])
for motion in (False, True):
    fn = lambda: ops.imu_preint(dt, gyro, acc, seg_d, seg_h, ip, ir, iv, tr['gravity'], motion)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    print('imu_preint 5000 frames, %s mode: %.1f us per call' % ('motion' if motion else 'world', a.elapsed_time(b) / 20 * 1e3))
fn = lambda: ops.imu_preint_both(dt, gyro, acc, seg_d, seg_h, ip, ir, iv, tr['gravity'])
for _ in range(3): fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): fn()
b.record(); torch.cuda.synchronize()
print('imu_preint_both 5000 frames (world + motion rows from one pass): %.1f us per call' % (a.elapsed_time(b) / 20 * 1e3))
from torch.profiler import profile, ProfilerActivity
import collections
for motion in (False, True):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            ops.imu_preint(dt, gyro, acc, seg_d, seg_h, ip, ir, iv, tr['gravity'], motion)
        torch.cuda.synchronize()
    acc_ = collections.defaultdict(float)
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            acc_[e.name[:60]] += e.device_time / 5
    print('%s mode kernels (us): ' % ('motion' if motion else 'world') + ', '.join('%s %.0f' % (k.split('(')[0].split('::')[-1], v) for k, v in sorted(acc_.items(), key=lambda kv: -kv[1])[:6]))
