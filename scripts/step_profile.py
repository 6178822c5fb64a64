"""GPU-kernel breakdown of the whole bilevel step (bench.py stereo_vio set-up, sequential schedule), grouped by kernel name."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from islam_amd import lietensor as pp, synthetic
from islam_amd.TartanVO import TartanVO
from islam_amd.bilevel import BilevelLoop
from islam_amd.imu_integrator import IMUModule
device = torch.device('cuda:0')
batch, steps, warmup = 8, 3, 3
torch.manual_seed(0)
vo = TartanVO(correct_scale=False, fix_parts=("flow", "stereo"), use_kitti_coord=True, frozen_dtype=torch.bfloat16,
              flow_dtype=torch.bfloat16, host_glue=True, miopen_find=os.environ.get('FIND', '1') == '1',
              pose_channels_last=os.environ.get('POSE_CL', '1') == '1', graph_frozen=os.environ.get('GRAPH', '1') == '1')
with torch.no_grad():
    vo.vonet.stereoNet.conv_c13.weight.zero_(); vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
n = (steps + warmup) * batch + 1
tr = synthetic.car_trajectory(n, seed=3)
imu = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], np.zeros(3), np.zeros(3), tr['init'], tr['gravity'],
                tr['rgb2imu_sync'], device=str(device), denoise_model_name=None, denoise_accel=True, denoise_gyro=False)
loop = BilevelLoop(vo, imu, pp.identity_SE3(), tr['init'], batch_size=batch, device=str(device))
samples = []
for k in range(2):
    smp = synthetic.stereo_batch(batch, seed=50 + k)
    samples.append({kk: (v.to(device) if isinstance(v, torch.Tensor) and (kk.startswith('img') or kk == 'intrinsic') else v)
                    for kk, v in smp.items()})
seq = []
for k in range(steps + warmup + 1):
    smp = dict(samples[k % 2]); smp['link'] = samples[k % 2]['link'] + k * batch; seq.append(smp)
loop.reset()
for k in range(warmup):
    loop.step(seq[k])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for k in range(warmup, warmup + steps):
        loop.step(seq[k])
    torch.cuda.synchronize()
rows, tot = [], 0.0
for e in prof.key_averages():
    t = getattr(e, 'self_device_time_total', None) or getattr(e, 'self_cuda_time_total', 0)
    if t > 0:
        rows.append((t / steps / 1e3, e.count // steps, e.key)); tot += t / steps / 1e3
rows.sort(reverse=True)
print('total GPU ms per step: %.2f' % tot)
for t, n_, k in rows[:int(os.environ.get('TOP', 40))]:
    print('%7.3f ms  n=%-4d %s' % (t, n_, k[:120]))
