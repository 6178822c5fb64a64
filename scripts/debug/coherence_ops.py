"""Which single kernel family, looping on a side stream, makes islam_scale_ls (main stream, inputs long in memory) return a different
mask from one launch to the next?  LOAD=<name>; see scripts/debug/coherence_bisect.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from islam_amd import ops
dev = torch.device('cuda:0')
load = os.environ.get('LOAD', 'conv_nhwc')
iters = int(os.environ.get('ITERS', '200'))
reps = int(os.environ.get('REPS', '20'))
B, H, W = 8, 112, 160
g = torch.Generator(device=dev).manual_seed(0)
side = torch.cuda.Stream(dev)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
cl = lambda t: t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
if load == 'conv_nhwc':
    x = cl(rn(16, 128, 112, 160)); w = ops.pack_conv_nhwc_weight(rn(128, 128, 3, 3) / 30)
    one = lambda: ops.conv_nhwc(x, w, 128, 3)
elif load == 'conv_nhwc32':
    x = cl(rn(16, 32, 224, 320)); w = ops.pack_conv_nhwc_weight(rn(32, 32, 3, 3) / 17)
    one = lambda: ops.conv_nhwc(x, w, 32, 3, stats=True)
elif load == 'conv_nhwc_flow':
    xm = cl(rn(8, 256, 112, 160)); w = ops.pack_conv_nhwc_weight(rn(128, 256, 3, 3) / 40); bias = rn(128)
    y32 = torch.empty(8, 128, 112, 160, device=dev); ym = cl(torch.empty(8, 128, 112, 160, device=dev))
    one = lambda: ops.conv_nhwc_flow(xm, 0, 256, w, bias, y32, 0, 128, 0.1, ymir=ym, moff=0)
elif load == 'conv3x3_mfma':
    x = rn(8, 64, 56, 80); w = ops.pack_conv3x3_weight(rn(96, 64, 3, 3) / 24); bias = rn(96)
    one = lambda: ops.conv3x3_mfma(x, w, bias, 96, stride=2)
elif load == 'corr81':
    f1, f2 = rn(8, 32, 112, 160), rn(8, 32, 112, 160); buf = torch.empty(8, 89, 112, 160, device=dev)
    one = lambda: ops.corr81_act(f1, f2, buf, 8, 0.1)
elif load == 'warp':
    f2 = rn(8, 32, 112, 160); fl = rn(8, 2, 112, 160)
    one = lambda: ops.warp_mask(f2, fl, 5.0)
elif load == 'pyramid':
    xi = rn(16, 3, 448, 640)
    ws = [ops.pack_pyramid_weight(rn(16, c, 3, 3) * (2.0 / (9 * c)) ** 0.5) for c in (3, 16, 16)]; bs = [rn(16) * 0.1 for _ in range(3)]
    one = lambda: ops.flow_pyramid_level(xi, ws, bs, 0.1)
elif load == 'mirror':
    src = rn(8, 128, 112, 160); dst = cl(torch.empty(8, 128, 112, 160, device=dev))
    one = lambda: ops.nchw_to_nhwc_mirror(src, 0, 128, dst, 0)
elif load == 'bn_apply':
    x = cl(rn(16, 128, 112, 160)); ss = rn(256)
    one = lambda: ops.bn_apply_(x, ss, relu=True)
elif load == 'torch_conv':
    x = cl(rn(16, 128, 112, 160)); w = cl(rn(128, 128, 3, 3) / 30)
    one = lambda: torch.nn.functional.conv2d(x, w, padding=1)
elif load == 'torch_elem':
    x = rn(16, 128, 112, 160)
    one = lambda: x * 1.0001
else:
    one = lambda: None
one(); torch.cuda.synchronize()

disp = torch.full((B, 1, H, W), 10.0, device=dev)
flow = torch.randn(B, 2, H, W, device=dev, generator=g)
pose7 = torch.tensor([[0.1, 0.2, 1.0, 0, 0, 0, 1.0]] * B, device=dev)
intr4 = torch.tensor([[180.0, 180.0, 80.0, 56.0]] * B).to(dev)
baseline = torch.full((B,), 0.5).to(dev)
th = torch.full((B,), 5.0).to(dev)
u = (torch.rand(B, H, W, device=dev, generator=g) > 0.5).to(torch.uint8)
torch.cuda.synchronize()
want = ops.scale_ls(disp, flow, pose7, intr4, baseline, u, th)[2].clone()
torch.cuda.synchronize()
victims = {
    'scale_ls': lambda: ops.scale_ls(disp, flow, pose7, intr4, baseline, u, th)[2],
    'torch_mask': lambda: ((flow * flow).sum(1) > 0) & (u != 0) & (disp[:, 0] >= 5.0),
    'torch_sum': lambda: (flow.abs().sum(1) * u).sum(dim=(0, 1)),
    'warp_mask': lambda: ops.warp_mask(f2v, flow, 1.0),
    'corr81': lambda: ops.corr81_forward(f1v, f2v),
}
f1v, f2v = rn(8, 32, 112, 160), rn(8, 32, 112, 160)
wants = {}
for k, f in victims.items():
    wants[k] = f().clone()
    torch.cuda.synchronize()
bad = {k: 0 for k in victims}
for it in range(iters):
    for k, f in victims.items():
        with torch.cuda.stream(side):
            for _ in range(reps):
                one()
        rs = [f() for _ in range(4)]
        torch.cuda.synchronize()
        bad[k] += sum(int(not torch.equal(r, wants[k])) for r in rs)
print('load=%-16s wrong results of %d launches per victim: %s' % (load, 4 * iters, bad))
