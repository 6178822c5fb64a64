"""The pose head's backward gives, in a few percent of its runs and WITHOUT any concurrent load, gradients 1e-4 .. 7e-3 off
(scripts/debug/thirdparty_stability.py).  Which tensor deviates first?  Tensor hooks on every stage's output record its gradient; on a glitch
run the hooks are compared with a clean run, from the loss backwards."""
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
vn.set_pose_channels_last(os.environ.get('CL', '1') == '1')
if os.environ.get('FUSED_TAIL') == '0':
    head.set_fused_tail(False)
B = 8
x0 = torch.cat([torch.randn(B, 2, 112, 160), torch.rand(B, 2, 112, 160)], 1).to(dev)
if os.environ.get('CL', '1') == '1':
    x0 = x0.contiguous(memory_format=torch.channels_last)
wts = torch.arange(1, 7, device=dev, dtype=torch.float32)
grads = {}


def tap(name):
    def hook(g):
        grads[name] = g.detach().clone()
    return hook


def run():
    grads.clear()
    x = x0.clone().requires_grad_(True)
    h = x
    fused = getattr(head, 'fused_tail', False) and nets._fused_tail_ok(h)
    for i, m in enumerate(head.feat_net):
        if i < 3:
            h = nets._conv_relu_fused(m, h) if fused else m(h)
            h.register_hook(tap('feat_net.%d' % i))
        else:
            for j, blk in enumerate(m):
                h = blk(h)
                h.register_hook(tap('feat_net.%d.%d' % (i, j)))
    f = h.reshape(h.shape[0], -1)
    f.register_hook(tap('flatten'))
    y = torch.cat((head.voflow_trans(f), head.voflow_rot(f)), 1)
    (y * wts).sum().backward()
    return dict(grads), x.grad.detach().clone()


rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for _ in range(10):
    ref, refx = run()
torch.cuda.synchronize()
found = 0
for it in range(int(os.environ.get('N', '400'))):
    g, gx = run()
    torch.cuda.synchronize()
    r = rel(gx, refx)
    if r > 1e-5:
        found += 1
        order = list(ref.keys())                                  # hooks fire in backward order: from the loss towards the input
        devs = [(k, rel(g[k], ref[k])) for k in order]
        first = next((k for k, d in devs if d > 1e-5), None)
        print('run %d: input gradient off by %.1e; from the loss backwards: %s' % (it, r, ' '.join('%s=%.0e' % (k.replace('feat_net.', ''), d) for k, d in devs)))
        print('   first deviating tensor: %s' % first)
        if found >= 4:
            break
print('%d glitch runs' % found)
