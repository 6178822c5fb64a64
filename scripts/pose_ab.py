"""Where the trainable pose head's time goes, and what MIOpen's kernel choices cost: forward + backward of VOFlowRes at B=8
(channels_last, HIP graphs off so that the profiler sees kernels), GPU time per step by kernel; DET=1 sets
torch.backends.cudnn.deterministic (no split-K atomics kernels -> no zero-fill / cast helper launches)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from islam_amd import nets
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
if os.environ.get('DET') == '1':
    torch.backends.cudnn.deterministic = True
torch.manual_seed(0)
head = nets.VOFlowRes(fix_parts=('flow', 'stereo')).to(dev).train().to(memory_format=torch.channels_last)
x = torch.randn(8, 4, 112, 160, device=dev).contiguous(memory_format=torch.channels_last)


def step():
    for p in head.parameters():
        p.grad = None
    (head(x) ** 2).sum().backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    step()
b.record()
torch.cuda.synchronize()
print('DET=%s: pose head fwd+bwd %.3f ms per step (eager, wall on the stream)' % (os.environ.get('DET', '0'), a.elapsed_time(b) / 10))
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        acc[e.name][0] += e.device_time / 3e3
        acc[e.name][1] += 1
print('GPU ms per step %.3f, launches %d' % (sum(v[0] for v in acc.values()), sum(v[1] for v in acc.values()) // 3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:22]:
    print('  %7.3f ms  n=%-4d %s' % (v[0], v[1] // 3, k[:110]))
