"""Which op of the bf16 channels-last stereo execution copy rounds fp32 -> bf16 with something other than round-to-nearest-even?
For each op: fraction of outputs that differ from RNE(fp32 reference on the same bf16 operands), and how many of those lie
TOWARDS ZERO of the reference (truncation) -- a systematic magnitude loss of ~2^-9 per layer.
GPU box:  python scripts/calib/bf16_rounding_probe.py"""
import torch
import torch.nn.functional as F

dev = torch.device('cuda:0')
torch.manual_seed(0)
cl = torch.channels_last


def report(name, got, ref32):
    want = ref32.to(torch.bfloat16)
    g, w, r = got.float().flatten(), want.float().flatten(), ref32.float().flatten()
    diff = g != w
    toward_zero = (g.abs() < w.abs()) & diff
    ulp_err = ((g - r).abs() / (r.abs().clamp_min(1e-30) * 2.0 ** -8))
    print('%-46s differs from RNE: %6.2f %%   of those towards zero: %5.1f %%   mean signed rel err*2^9: %+.3f   max err %.2f ulp' % (
        name, 100 * diff.float().mean().item(), 100 * toward_zero.float().sum().item() / max(diff.sum().item(), 1),
        (((g.abs() - r.abs()) / r.abs().clamp_min(1e-30)).mean() * 512).item(), ulp_err.max().item()))


for (cin, cout, k, s, hw, B) in ((3, 32, 3, 2, (256, 256), 4), (32, 32, 3, 1, (128, 128), 4), (128, 128, 3, 1, (64, 64), 4), (320, 128, 3, 1, (64, 64), 2),
                                 (64, 64, 1, 1, (128, 128), 2), (32, 32, 3, 1, (64, 64), 2), (134, 64, 3, 1, (128, 128), 2), (256, 384, 3, 1, (16, 16), 2)):
    x = torch.randn(B, cin, *hw, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)
    w = (torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5).to(torch.bfloat16).contiguous(memory_format=cl)
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        y = F.conv2d(x, w, None, s, k // 2)
        ref = F.conv2d(x.float(), w.float(), None, s, k // 2)
        report('conv %d->%d k%d s%d %s find=%s' % (cin, cout, k, s, hw, bench), y, ref)
torch.backends.cudnn.benchmark = False
for (cin, cout, hw) in ((896, 320, (8, 8)), (128, 64, (128, 128))):
    x = torch.randn(2, cin, *hw, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)
    w = (torch.randn(cin, cout, 4, 4, device=dev) / (cin * 4) ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, device=dev).to(torch.bfloat16)
    y = F.conv_transpose2d(x, w, b, 2, 1)
    report('deconv %d->%d %s' % (cin, cout, hw), y, F.conv_transpose2d(x.float(), w.float(), b.float(), 2, 1))
x = torch.randn(2, 64, 128, 128, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)
report('max_pool2d', F.max_pool2d(x, 2), F.max_pool2d(x.float(), 2))
report('interpolate 0.5 bilinear', F.interpolate(x, scale_factor=0.5, mode='bilinear'), F.interpolate(x.float(), scale_factor=0.5, mode='bilinear'))
report('block mean 8', x.reshape(2, 64, 16, 8, 16, 8).mean((3, 5)), x.float().reshape(2, 64, 16, 8, 16, 8).mean((3, 5)))
report('relu', F.relu(x), F.relu(x.float()))
report('add', x + x.flip(0), x.float() + x.float().flip(0))
w = (torch.randn(64, 64, 1, 1, device=dev) / 8).to(torch.bfloat16)
b = torch.randn(64, device=dev).to(torch.bfloat16)
report('conv1x1 + bias (nn.Conv2d path)', F.conv2d(x, w, b), F.conv2d(x.float(), w.float(), b.float()))
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from islam_amd import ops  # noqa: E402
report('HIP resize', ops.resize_bilinear(x, (256, 256), align_corners=False), F.interpolate(x.float(), (256, 256), mode='bilinear', align_corners=False))
