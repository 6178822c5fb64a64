"""Forward / backward time of the trainable pose head at the benched shape (B = 8, 4 x 112 x 160): the hand-written kernels of
csrc/pose_head.hip (direct calls and HIP-graph replay) against the torch modules on MIOpen / CK (channels-last, fused tail, the pinned
solution set) as two captured graphs (nets._PoseGraph).  Usage: python scripts/pose_head_bench.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import nets, pose_head
from islam_amd.miopen_pin import use_pinned_db

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
use_pinned_db()
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = nets.VOFlowRes().to(dev).to(memory_format=torch.channels_last)
x = torch.randn(8, 4, 112, 160, device=dev).contiguous(memory_format=torch.channels_last)
gy = torch.randn(8, 6, device=dev)


def timed(fn, n=reps):
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


for graphs in (False, True):
    h = pose_head.PoseHeadHip(net, graphs=graphs)
    h._prepare(x)
    with torch.no_grad():
        f = timed(lambda: h.forward_raw(x))
    h.forward_raw(x)
    b = timed(lambda: h.backward_raw(gy))
    print('hip head, graphs=%d: forward %.1f us, backward (+ accumulate) %.1f us' % (graphs, f, b), flush=True)
    for p in net.parameters():
        p.grad = None
if os.environ.get('SKIP_TORCH') != '1':
    net.set_fused_tail(True)
    pg = nets._PoseGraph(net, x)
    f = timed(lambda: pg.fwd.replay())
    b = timed(lambda: pg.bwd.replay())
    print('torch modules (MIOpen / CK + fused tail), graph replay: forward %.1f us, backward %.1f us' % (f, b), flush=True)
