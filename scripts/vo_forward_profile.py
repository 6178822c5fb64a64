"""GPU-kernel breakdown of the TartanVO forward at B=8 (bench.py stereo_vio shapes), grouped by kernel name."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from islam_amd import synthetic
from islam_amd.TartanVO import TartanVO
dev = torch.device('cuda:0')
B = 8
if os.environ.get('BENCHMARK') == '1':          # MIOpen exhaustive find instead of its heuristics / find-db
    torch.backends.cudnn.benchmark = True
TOP = int(os.environ.get('TOP', 45))
torch.manual_seed(0)
vo = TartanVO(correct_scale=False, fix_parts=("flow", "stereo"), use_kitti_coord=True, frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16)
with torch.no_grad():
    vo.vonet.stereoNet.conv_c13.weight.zero_(); vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
s = synthetic.stereo_batch(B, seed=100)
s = {kk: (v.to(dev) if isinstance(v, torch.Tensor) and (kk.startswith('img') or kk == 'intrinsic') else v) for kk, v in s.items()}
for _ in range(3):
    vo(s)
torch.cuda.synchronize()
reps = 3
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(reps):
        vo(s)
    torch.cuda.synchronize()
tot = 0.0
rows = []
for e in prof.key_averages():
    t = getattr(e, 'self_device_time_total', None) or getattr(e, 'self_cuda_time_total', 0)
    if t > 0:
        rows.append((t / reps / 1e3, e.count // reps, e.key))
        tot += t / reps / 1e3
rows.sort(reverse=True)
print('total GPU ms per forward: %.2f' % tot)
for t, n, k in rows[:TOP]:
    print('%7.3f ms  n=%-4d %s' % (t, n, k[:110]))
