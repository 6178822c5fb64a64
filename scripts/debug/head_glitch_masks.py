"""Follow-up of head_glitch.py: are the rare "glitch" gradients of the pose head a flipped ReLU mask?  MIOpen's split-K kernels add their
partial sums with atomics, so a forward activation differs by ~1e-7 relative between runs; an element whose pre-activation sits that
close to zero changes the sign of its ReLU, and its whole upstream gradient appears / disappears -- a fixed-size, reproducible step.
Runs the head with the unfused tail (plain torch modules, hooks on every ReLU output) and, on a run whose input gradient deviates,
lists the ReLU outputs whose >0 mask differs from the reference run and the size of the activations there."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from islam_amd import nets
from islam_amd.miopen_pin import use_pinned_db
use_pinned_db()
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')
torch.manual_seed(3)
vn = nets.VONet(fix_parts=('flow', 'stereo'))
head = vn.flowPoseNet.to(dev).train()
vn.set_pose_channels_last(True)
head.set_fused_tail(False)
B = 8
x0 = torch.cat([torch.randn(B, 2, 112, 160), torch.rand(B, 2, 112, 160)], 1).to(dev).contiguous(memory_format=torch.channels_last)
wts = torch.arange(1, 7, device=dev, dtype=torch.float32)
acts = {}


def keep(name):
    def hook(_m, _i, out):
        acts[name] = out.detach().clone()
    return hook


for i, m in enumerate(head.feat_net):
    if i < 3:
        m.register_forward_hook(keep('feat_net.%d' % i))
    else:
        for j, blk in enumerate(m):
            blk.conv1.register_forward_hook(keep('feat_net.%d.%d.conv1' % (i, j)))
            blk.register_forward_hook(keep('feat_net.%d.%d' % (i, j)))


def run():
    acts.clear()
    x = x0.clone().requires_grad_(True)
    y = head(x)
    y = torch.cat(y, 1) if isinstance(y, (tuple, list)) else y
    (y * wts).sum().backward()
    return dict(acts), x.grad.detach().clone()


rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for _ in range(10):
    ref, refx = run()
torch.cuda.synchronize()
found = noisy = 0
for it in range(int(os.environ.get('N', '800'))):
    a, gx = run()
    torch.cuda.synchronize()
    r = rel(gx, refx)
    fwd_diff = [k for k in ref if not torch.equal(a[k], ref[k])]
    noisy += bool(fwd_diff)
    if r > 1e-5:
        found += 1
        print('run %d: input gradient off by %.1e; %d of %d forward activations differ bitwise from the reference run' % (it, r, len(fwd_diff), len(ref)))
        for k in ref:
            flip = (a[k] > 0) != (ref[k] > 0)
            n = int(flip.sum())
            if n:
                print('   %-22s %d element(s) change ReLU sign; |activation| there: this run %.2e, reference run %.2e (tensor max %.2e)'
                      % (k, n, float(a[k][flip].abs().max()), float(ref[k][flip].abs().max()), float(ref[k].abs().max())))
        if found >= 6:
            break
print('%d glitch runs; %d of %d runs have a forward activation that differs bitwise from the reference run' % (found, noisy, it + 1))
