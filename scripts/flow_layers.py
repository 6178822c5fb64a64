"""GPU kernels of ONE forward of the frozen flow net in launch order (B=8, 448x640): name, grid, duration -- which layers the time sits in."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from islam_amd import nets
from islam_amd.miopen_pin import use_pinned_db
use_pinned_db()
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
vonet = nets.VONet(fix_parts=('flow', 'stereo')).to(dev).train()
vonet.set_frozen_dtype(torch.bfloat16, torch.bfloat16)
x = torch.randn(8, 6, 448, 640, device=dev)
run = lambda: vonet._run_frozen('flow', vonet.flowNet, torch.bfloat16, x)
with torch.no_grad():
    for _ in range(4):
        run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run()
        torch.cuda.synchronize()
ev = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA), key=lambda e: e.time_range.start)
t0 = ev[0].time_range.start
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for e in ev[:n]:
    print('%8.1f us  +%7.1f  %s' % (e.time_range.start - t0, e.device_time, e.name[:90]))
print('span %.1f us, kernel sum %.1f us' % (ev[-1].time_range.end - t0, sum(e.device_time for e in ev)))
