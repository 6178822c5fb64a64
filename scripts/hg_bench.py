"""Hourglass Residual modules of the stereo net at B=8: fused launch (islam_hg_residual_nhwc_bf16) against the three-launch path,
per (channels, map) shape of StereoNet7 (448x640 input).  us per module, burst of launches between one event pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import nets
dev = torch.device('cuda:0')
torch.manual_seed(0)
SHAPES = [(64, 64, 224, 320), (64, 64, 112, 160), (64, 64, 56, 80),
          (128, 192, 112, 160), (192, 192, 56, 80), (192, 192, 28, 40),
          (192, 256, 56, 80), (256, 256, 28, 40), (256, 256, 14, 20),
          (192, 192, 14, 20), (128, 128, 112, 160), (128, 128, 56, 80), (128, 128, 28, 40)]
only = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else None


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for i, (cin, cout, H, W) in enumerate(SHAPES):
    if only is not None and i not in only:
        continue
    m = nets._HGResidual(cin, cout).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last).eval()
    x = torch.randn(8, cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    out = {}
    with torch.no_grad():
        for fused in (True, False):
            nets.HG_FUSED = fused
            g = torch.cuda.CUDAGraph()                       # graph replay: no host launch gaps, like the benched forward
            m(x); torch.cuda.synchronize()
            with torch.cuda.graph(g):
                for _ in range(10):
                    y = m(x)
            out[fused] = timed(g.replay, 20) / 10
    print('%3d->%3d @%3dx%3d  fused %6.1f us   layer-wise %6.1f us   (%.2fx)' % (cin, cout, H, W, out[True], out[False], out[False] / out[True]), flush=True)
