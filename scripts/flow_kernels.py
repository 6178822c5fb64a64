"""torch.profiler GPU-kernel breakdown of ONE forward of the frozen flow net (B=8, 448x640, bf16 operands on the HIP kernels)."""
import os, sys, collections
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
        for _ in range(3):
            run()
        torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        acc[e.name][0] += e.device_time / 3e3
        acc[e.name][1] += 1
tot = sum(v[0] for v in acc.values())
print('total GPU ms per flow forward: %.2f, launches %d' % (tot, sum(v[1] for v in acc.values()) // 3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:40]:
    print('  %7.3f ms  n=%-4d %s' % (v[0], v[1] // 3, k[:140]))
