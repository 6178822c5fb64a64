"""Fused pyramid level (islam_flow_pyramid_level) against the launch-per-layer path of PWCDCNet.forward_mfma, levels 1 and 2, B=8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from islam_amd import nets
from islam_amd.miopen_pin import use_pinned_db
use_pinned_db()
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
net = nets.PWCDCNet().to(dev).eval()
x1 = torch.randn(16, 3, 448, 640, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


with torch.no_grad():
    def layers(l, f):
        for s in ('a', 'aa', 'b'):
            f = net._c('conv%d%s' % (l, s), f)
        return f
    f1 = layers(1, x1)
    g1 = net._pyramid_level(1, x1)
    print('level 1: max |fused - layers| = %.3e of %.3e' % (float((f1 - g1).abs().max()), float(f1.abs().max())))
    f2 = layers(2, f1)
    g2 = net._pyramid_level(2, f1)
    print('level 2: max |fused - layers| = %.3e of %.3e' % (float((f2 - g2).abs().max()), float(f2.abs().max())))
    print('level 1: layers %.1f us   fused %.1f us' % (timed(lambda: layers(1, x1)), timed(lambda: net._pyramid_level(1, x1))))
    print('level 2: layers %.1f us   fused %.1f us' % (timed(lambda: layers(2, f1)), timed(lambda: net._pyramid_level(2, f1))))
