"""Can the frozen flow + disparity forward be captured in a HIP graph?  Outputs and wall time per call, eager vs replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import nets
dev = torch.device('cuda:0')
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
B = 8
vn = nets.VONet(fix_parts=('flow', 'stereo')).to(dev)
vn.set_frozen_dtype(torch.bfloat16, torch.bfloat16)
ins = [torch.rand(B, 3, 448, 640, device=dev) for _ in range(2)] + [torch.randn(B, 3, 448, 640, device=dev) for _ in range(2)]
def eager():
    with torch.no_grad():
        return vn.frozen_forward(*ins)
for _ in range(3): f0, d0 = eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): eager()
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 5
print('eager %.2f ms' % (te * 1e3), flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): eager()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    fo, do = eager()
torch.cuda.synchronize()
print('captured', flush=True)
g.replay(); torch.cuda.synchronize()
print('max |flow diff| %.3e  max |disp diff| %.3e' % ((fo - f0).abs().max().item(), (do - d0).abs().max().item()), flush=True)
t0 = time.perf_counter()
for _ in range(5): g.replay()
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 5
print('graph replay %.2f ms' % (tg * 1e3), flush=True)
