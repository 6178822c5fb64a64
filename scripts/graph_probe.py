import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import nets
dev = torch.device('cuda:0')
B = 8
torch.manual_seed(0)
vonet = nets.VONet(fix_parts=('flow', 'stereo')).to(dev)
vonet.set_frozen_dtype(torch.bfloat16)
vonet.train()
x_st = torch.randn(B, 6, 448, 640, device=dev).contiguous(memory_format=torch.channels_last)
x_fl = torch.rand(B, 6, 448, 640, device=dev)

def run_stereo():
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        return vonet.stereoNet(x_st)[0]
def run_flow():
    with torch.no_grad():
        return vonet.flowNet(x_fl)[0][0]

def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

print('eager stereo %.2f ms, flow %.2f ms' % (timeit(run_stereo), timeit(run_flow)), flush=True)
for name, fn in (('stereo', run_stereo), ('flow', run_flow)):
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): fn()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        t = timeit(g.replay)
        print('graph %s %.2f ms' % (name, t), flush=True)
    except Exception as e:
        print('graph %s FAILED: %s' % (name, str(e)[:300]), flush=True)
