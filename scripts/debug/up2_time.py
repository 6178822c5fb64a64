import os, sys
sys.path.insert(0, '/root/repo')
import torch
from islam_amd import nets, ops
from islam_amd.miopen_pin import use_pinned_db
use_pinned_db()
torch.backends.cudnn.benchmark = True
net = nets.PWCDCNet().cuda()
def t_us(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
tot_h = tot_m = 0
for l, (H, W) in ((6, (7, 10)), (5, (14, 20)), (4, (28, 40)), (3, (56, 80))):
    for name in ('deconv%d' % l, 'upfeat%d' % l):
        dc = getattr(net, name)
        x = torch.randn(8, dc.in_channels, H, W, device='cuda')
        h = t_us(lambda: ops.deconv_to2(x, dc.weight.detach(), dc.bias.detach()))
        with torch.no_grad():
            m = t_us(lambda: dc(x))
        tot_h += h; tot_m += m
        print('%-8s C=%3d %2dx%2d: HIP %6.1f us   MIOpen %6.1f us' % (name, dc.in_channels, H, W, h, m))
print('sum: HIP %.1f us, MIOpen %.1f us' % (tot_h, tot_m))
