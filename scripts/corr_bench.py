"""islam_corr81_fwd_act at the five PWC levels, B=8: us per call (burst between one event pair) and fraction of the 8 TB/s HBM
roof of the algorithmic bytes 4 B H W (2 C + 81).  ISLAM_CORR4=0 / 1 / 2 selects the kernel variant (one process each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import ops
dev = torch.device('cuda:0')
for (C, H, W) in [(196, 7, 10), (128, 14, 20), (96, 28, 40), (64, 56, 80), (32, 112, 160)]:
    f1 = torch.randn(8, C, H, W, device=dev); f2 = torch.randn(8, C, H, W, device=dev)
    buf = torch.empty(8, 81 + 8, H, W, device=dev)
    fn = lambda: ops.corr81_act(f1, f2, buf, 8, 0.1)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    by = 4.0 * 8 * H * W * (2 * C + 81)
    print('CORR4=%s  C=%3d %3dx%3d  %6.1f us  %5.2f TB/s = %.3f of 8 TB/s' % (os.environ.get('ISLAM_CORR4', '1'), C, H, W, us, by / us / 1e6, by / us / 8e6), flush=True)
