import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import ops
dev = torch.device('cuda:0')
B = 8
for C, H, W in [(196, 7, 10), (128, 14, 20), (96, 28, 40), (64, 56, 80), (32, 112, 160)]:
    f1, f2 = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
    fl = torch.randn(B, 2, H, W, device=dev)
    for _ in range(20):
        ops.corr81_forward(f1, f2)
        ops.warp_mask(f1, fl, 1.0)
torch.cuda.synchronize()
