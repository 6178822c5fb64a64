"""Is the pipelined bilevel step bound by the GPU or by the host?  Kernel intervals of six steps (torch profiler): the union of the
intervals (time with at least one kernel running) against the wall span, and the time with at least two kernels running."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from islam_amd import lietensor as pp, synthetic
from islam_amd.TartanVO import TartanVO
from islam_amd.bilevel import BilevelLoop
from islam_amd.imu_integrator import IMUModule
from islam_amd.miopen_pin import use_pinned_db
use_pinned_db()
device = torch.device('cuda:0')
batch = 8
torch.manual_seed(0)
vo = TartanVO(correct_scale=False, fix_parts=("flow", "stereo"), use_kitti_coord=True, frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16,
              host_glue=True, miopen_find=True, pose_channels_last=True, graph_frozen=True, graph_pose='accumulate')
with torch.no_grad():
    vo.vonet.stereoNet.conv_c13.weight.zero_(); vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
steps, warmup = 6, 4
tr = synthetic.car_trajectory((steps + warmup) * batch + 1, seed=3)
imu = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], np.zeros(3), np.zeros(3), tr['init'], tr['gravity'], tr['rgb2imu_sync'],
                device=str(device), denoise_model_name=None, denoise_accel=True, denoise_gyro=False)
loop = BilevelLoop(vo, imu, pp.identity_SE3(), tr['init'], batch_size=batch, device=str(device))
samples = []
for k in range(2):
    smp = synthetic.stereo_batch(batch, seed=50 + k)
    samples.append({kk: (v.to(device) if isinstance(v, torch.Tensor) and (kk.startswith('img') or kk == 'intrinsic') else v) for kk, v in smp.items()})
seq = []
for k in range(steps + warmup + 2):
    smp = dict(samples[k % 2]); smp['link'] = samples[k % 2]['link'] + k * batch; seq.append(smp)
loop.reset()
for k in range(warmup):
    loop.step(seq[k], next_sample=seq[k + 1])
torch.cuda.synchronize()
for k_ in loop.timing: loop.timing[k_] = 0.0
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for k in range(warmup, warmup + steps):
        loop.step(seq[k], next_sample=seq[k + 1])
    torch.cuda.synchronize()
print('host time per stage and step (pipelined schedule, ms): ' + ', '.join('%s %.2f' % (k_, v_ / steps * 1e3) for k_, v_ in loop.timing.items()))
# busy time per stream (the union of a stream's kernel intervals)
import collections
per = collections.defaultdict(list)
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        per[e.thread].append((e.time_range.start, e.time_range.end))
for sid, ivs in sorted(per.items(), key=lambda kv: -len(kv[1])):
    ivs.sort()
    tot, end = 0.0, None
    for a, b in ivs:
        if end is None or a > end:
            tot += b - a; end = b
        elif b > end:
            tot += b - end; end = b
    print('  stream %s: %d kernels per step, busy %.2f ms per step, kernel-time sum %.2f ms per step' % (sid, len(ivs) // steps, tot / steps / 1e3, sum(b - a for a, b in ivs) / steps / 1e3))
iv = sorted((e.time_range.start, e.time_range.end) for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
t0, t1 = iv[0][0], max(b for _, b in iv)
pts = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
depth, last, busy1, busy2 = 0, t0, 0.0, 0.0
for t, d in pts:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
span = t1 - t0
print('span %.2f ms for %d steps (%.2f ms per step); >=1 kernel running %.1f %%, >=2 running %.1f %%; kernel-time sum %.2f ms per step; %d launches per step'
      % (span / 1e3, steps, span / steps / 1e3, 100 * busy1 / span, 100 * busy2 / span, sum(b - a for a, b in iv) / steps / 1e3, len(iv) // steps))
# idle gaps (no kernel running) longer than 40 us in the third profiled step, with the kernels on either side
ev = sorted(((e.time_range.start, e.time_range.end, e.name) for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA))
lo, hi = t0 + 2 * span / steps, t0 + 3 * span / steps
end, endname, rows = None, None, []
for a, b, n in ev:
    if end is not None and a - end > 40 and lo <= end < hi:
        rows.append((end - lo, a - end, endname[:50], n[:50]))
    if end is None or b > end:
        end, endname = b, n
print('idle gaps > 40 us inside one step (offset us, gap us, kernel before -> kernel after):')
for r in rows:
    print('  %8.0f  %6.0f   %s  ->  %s' % r)
print('sum of those gaps: %.0f us' % sum(r[1] for r in rows))
if os.environ.get('GAP_CONTEXT') == '1':          # the launches around every gap, with start offsets and durations
    for off, gap, _, _ in rows:
        g0 = lo + off
        print('--- gap at %.0f us (%.0f us)' % (off, gap))
        near = [e for e in ev if g0 - 400 <= e[0] <= g0 + gap + 300]
        for a, b, n in near[-40:]:
            print('   %9.0f  +%6.1f  %s' % (a - lo, b - a, n[:70]))
