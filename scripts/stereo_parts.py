"""GPU time of the stereo net's execution copy by top-level part (B=8, 448x640): where the 18 ms go."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import nets
dev = torch.device('cuda:0')
torch.manual_seed(0)
vonet = nets.VONet(fix_parts=('flow', 'stereo')).to(dev).train()
vonet.set_frozen_dtype(torch.bfloat16)
x = torch.randn(8, 6, 448, 640, device=dev)
run = lambda: vonet._run_frozen('stereo', vonet.stereoNet, torch.bfloat16, x)
for _ in range(3): run()
m = vonet._exec['stereo'].module()
acc = collections.defaultdict(float)
evs = []
def hook(name):
    def pre(mod, inp):
        e = torch.cuda.Event(enable_timing=True); e.record(); mod._e0 = e
    def post(mod, inp, out):
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((name, mod._e0, e))
    return pre, post
for name, child in m.named_children():
    pre, post = hook(name)
    child.register_forward_pre_hook(pre); child.register_forward_hook(post)
fe = m.feature_extraction
for name, child in fe.named_children():
    pre, post = hook('fe.' + name)
    child.register_forward_pre_hook(pre); child.register_forward_hook(post)
reps = 5
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): run()
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / reps * 1e3
for name, a, b in evs:
    acc[name] += a.elapsed_time(b) / reps
print('stereo execution copy: %.2f ms per forward (wall, incl. launch gaps)' % tot)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('  %-22s %6.2f ms' % (k, v))
