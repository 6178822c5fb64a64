"""islam_warp_mask at the four PWC levels it is called on, B = 8: us per call (burst between one event pair) and fraction of the 8 TB/s
HBM roof of the algorithmic bytes 4 B H W (2 C + 2).  ISLAM_HIP_LIB selects the build (A/B runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import ops
dev = torch.device('cuda:0')
for (C, H, W), sc in zip([(128, 14, 20), (96, 28, 40), (64, 56, 80), (32, 112, 160)], (0.625, 1.25, 2.5, 5.0)):
    g = torch.Generator(device=dev).manual_seed(C)
    x = torch.randn(8, C, H, W, device=dev, generator=g)
    # a smooth field of a few pixels (what an optical-flow estimate looks like; bench.py's), not per-pixel noise
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    fl = torch.stack([0.6 * torch.sin(yy / 17.0) + 0.3 * torch.cos(xx / 23.0), 0.5 * torch.cos(yy / 29.0 + xx / 31.0)], 0)
    fl = (fl[None].repeat(8, 1, 1, 1).to(dev) + 0.02 * torch.randn(8, 2, H, W, device=dev, generator=g)).contiguous()
    fn = lambda: ops.warp_mask(x, fl, sc)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    by = 4.0 * 8 * H * W * (2 * C + 2)
    print('C=%3d %3dx%3d  %6.1f us  %5.2f TB/s = %.3f of 8 TB/s   checksum %.6f' % (C, H, W, us, by / us / 1e6, by / us / 8e6, float(fn().double().sum())), flush=True)
