"""torch.profiler table of one bilevel step (bench.py stereo_vio workload): where the 66 ms per batch go."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from islam_amd import lietensor as pp, synthetic
from islam_amd.TartanVO import TartanVO
from islam_amd.bilevel import BilevelLoop
from islam_amd.imu_integrator import IMUModule
dev = torch.device('cuda:0')
B = 8
torch.manual_seed(0)
vo = TartanVO(correct_scale=False, fix_parts=('flow', 'stereo'), use_kitti_coord=True, frozen_dtype=torch.bfloat16)
with torch.no_grad():
    vo.vonet.stereoNet.conv_c13.weight.zero_(); vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
tr = synthetic.car_trajectory(B * 8 + 1, seed=9)
imu = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], np.zeros(3), np.zeros(3), tr['init'], tr['gravity'], tr['rgb2imu_sync'],
                device='cuda:0', denoise_model_name=None, denoise_accel=True, denoise_gyro=False)
loop = BilevelLoop(vo, imu, pp.identity_SE3(), tr['init'], batch_size=B)
samples = []
for k in range(6):
    s = synthetic.stereo_batch(B, seed=100 + k)
    s['link'] = s['link'] + k * B
    samples.append({kk: (v.to(dev) if isinstance(v, torch.Tensor) and (kk.startswith('img') or kk == 'intrinsic') else v) for kk, v in s.items()})
for k in range(3):
    loop.step(samples[k])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for k in range(3, 6):
        loop.step(samples[k])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=28, max_name_column_width=70))
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=14, max_name_column_width=70))
print({k: v / 6 * 1e3 for k, v in loop.timing.items()})
