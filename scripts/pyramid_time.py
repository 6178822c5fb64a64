"""GPU time of the PWC feature pyramid alone (18 convolutions on both images as one batch of 2B; PWCNet.py:240-251) inside
PWCDCNet.forward_mfma, and of the whole flow forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import nets
from islam_amd.miopen_pin import use_pinned_db
use_pinned_db()
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = nets.PWCDCNet().to(dev).eval()
B = 8
x = torch.rand(B, 6, 448, 640, device=dev)


def pyramid(f):
    feats = []
    for l in range(1, 7):
        for s in (('a', 'aa', 'b') if l < 6 else ('aa', 'a', 'b')):
            f = net._c('conv%d%s' % (l, s), f)
        feats.append(f)
    return feats


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    f0 = torch.cat((x[:, 0:3], x[:, 3:6]), 0).contiguous()
    print('pyramid (2B = %d images): %.3f ms' % (2 * B, timeit(lambda: pyramid(f0))))
    print('whole flow forward      : %.3f ms' % timeit(lambda: net.forward_mfma(x)))
    g = torch.cuda.CUDAGraph()
    net.forward_mfma(x)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        net.forward_mfma(x)
    print('whole flow forward, graph replay: %.3f ms' % timeit(g.replay))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        pyramid(f0)
        torch.cuda.synchronize()
    rows = sorted(((getattr(e, 'self_device_time_total', 0) or 0) / 1e3, e.count, e.key) for e in prof.key_averages())
    for t, n, k in rows[::-1][:12]:
        if t > 0:
            print('%7.3f ms n=%-3d %s' % (t, n, k[:110]))
