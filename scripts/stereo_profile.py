import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import nets
dev = torch.device('cuda:0')
torch.manual_seed(0)
vonet = nets.VONet(fix_parts=('flow', 'stereo')).to(dev)
vonet.set_frozen_dtype(torch.bfloat16)
vonet.train()
x_st = torch.randn(8, 6, 448, 640, device=dev).contiguous(memory_format=torch.channels_last)
x_fl = torch.rand(8, 6, 448, 640, device=dev)
with torch.no_grad():
    for _ in range(6):
        with torch.autocast('cuda', dtype=torch.bfloat16):
            vonet.stereoNet(x_st)
        vonet.flowNet(x_fl)
torch.cuda.synchronize()
