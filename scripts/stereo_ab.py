"""A/B of the stereo execution copy (B=8, 448x640, train-mode BatchNorm): forward time with the convolutions on MIOpen only
(ISLAM_HIP_CONV=0) / hand-written 3x3 (=1) / + fused 1x1 (=2); run once per setting (the level is read at import)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import nets
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
vonet = nets.VONet(fix_parts=('flow', 'stereo')).to(dev).train()
vonet.set_frozen_dtype(torch.bfloat16)
x = torch.randn(8, 6, 448, 640, device=dev)
run = lambda: vonet._run_frozen('stereo', vonet.stereoNet, torch.bfloat16, x, quarter=os.environ.get('FULLRES') != '1')
with torch.no_grad():
    for _ in range(4):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
print('ISLAM_HIP_CONV=%s max_c=%s: stereo forward %.3f ms (graph replay, GPU time)' % (nets.HIP_CONV_LEVEL, nets.HIP_CONV_MAX_C, a.elapsed_time(b) / 10))
