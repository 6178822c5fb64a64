"""Flow net (PWC-Net forward_mfma) alone at B=8, 448x640: GPU time per forward and the kernel breakdown.
ISLAM_FLOW_NHWC=0 / 1 (read at import): fp32 NCHW convolution kernel / bf16 channels-last mirror."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from islam_amd import nets
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = nets.PWCDCNet().to(dev).eval()
x = torch.rand(8, 6, 448, 640, device=dev)
with torch.no_grad():
    for _ in range(3):
        net.forward_mfma(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        net.forward_mfma(x)
    b.record()
    torch.cuda.synchronize()
    print('ISLAM_FLOW_NHWC=%d: flow forward %.3f ms (eager, wall on the stream)' % (nets.FLOW_NHWC, a.elapsed_time(b) / 10))
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            net.forward_mfma(x)
        torch.cuda.synchronize()
rows, tot = [], 0.0
for e in prof.key_averages():
    t = getattr(e, 'self_device_time_total', None) or getattr(e, 'self_cuda_time_total', 0)
    if t > 0:
        rows.append((t / 3e3, e.count // 3, e.key)); tot += t / 3e3
rows.sort(reverse=True)
print('GPU ms per forward %.3f, launches %d' % (tot, sum(r[1] for r in rows)))
for t, n, k in rows[:int(os.environ.get('TOP', 14))]:
    print('  %7.3f ms  n=%-4d %s' % (t, n, k[:110]))
