"""Which backward kernel of the pose head deviates beside the conv_nhwc load (scripts/debug/thirdparty_stability.py)?  Per op: largest
deviation from an unloaded reference, relative to the tensor's size, without / with the load."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from islam_amd import ops
from islam_amd.miopen_pin import use_pinned_db
use_pinned_db()
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
xa = cl(rn(16, 128, 112, 160).to(torch.bfloat16)); wa = ops.pack_conv_nhwc_weight(rn(128, 128, 3, 3) / 30)
side = torch.cuda.Stream(dev)
cases = {}
for (B, C, H, W, Co, k, s) in ((8, 128, 14, 20, 128, 3, 1), (8, 64, 28, 40, 64, 3, 1), (8, 256, 4, 5, 256, 3, 1), (8, 32, 56, 80, 32, 3, 1), (8, 64, 28, 40, 128, 3, 2)):
    x = cl(rn(B, C, H, W)); w = cl(rn(Co, C, k, k) / (9 * C) ** 0.5)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    gy = cl(rn(B, Co, Ho, Wo))
    name = 'conv %d->%d @%dx%d s%d' % (C, Co, H, W, s)
    cases[name + ' fwd'] = (lambda x=x, w=w, s=s, k=k: F.conv2d(x, w, None, s, k // 2))
    cases[name + ' dgrad'] = (lambda x=x, w=w, gy=gy, s=s, k=k: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [k // 2, k // 2], [1, 1], False, [0, 0], 1, [True, False, False])[0])
    cases[name + ' wgrad'] = (lambda x=x, w=w, gy=gy, s=s, k=k: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [k // 2, k // 2], [1, 1], False, [0, 0], 1, [False, True, False])[1])
xb = cl(rn(8, 128, 14, 20)); bias = rn(128); res = cl(rn(8, 128, 14, 20)); gyb = cl(rn(8, 128, 14, 20))
def bias_bwd():
    x = xb.clone().requires_grad_(True); b = bias.clone().requires_grad_(True); r = res.clone().requires_grad_(True)
    ops.bias_act(x, b, r, True).backward(gyb)
    return torch.cat([x.grad.reshape(-1), b.grad.reshape(-1), r.grad.reshape(-1)])
cases['islam bias_act backward'] = bias_bwd
lin_w = rn(128, 1536) / 40; lin_x = rn(8, 1536); lin_g = rn(8, 128)
cases['linear 1536->128 wgrad (GEMM)'] = lambda: lin_g.t() @ lin_x
cases['linear 1536->128 dgrad (GEMM)'] = lambda: lin_g @ lin_w
ps = [rn(256, 256, 3, 3) for _ in range(20)]; gs = [rn(256, 256, 3, 3) for _ in range(20)]
cases['torch _foreach_add_'] = lambda: torch.cat([t.reshape(-1) for t in torch._foreach_add([p.clone() for p in ps], gs)])
rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for name, fn in cases.items():
    for _ in range(4):
        ref = fn().clone()
    torch.cuda.synchronize()
    dev_ = []
    for load in (False, True):
        worst = 0.0
        for i in range(30):
            if load:
                with torch.cuda.stream(side):
                    for _ in range(40):
                        ops.conv_nhwc(xa, wa, 128, 3)
            outs = [fn() for _ in range(3)]
            torch.cuda.synchronize()
            worst = max([worst] + [rel(o, ref) for o in outs])
        dev_.append(worst)
    print('%-34s unloaded %.2e   beside conv_nhwc %.2e%s' % (name, dev_[0], dev_[1], '   <-- VICTIM' if dev_[1] > 100 * max(dev_[0], 1e-7) else ''))
