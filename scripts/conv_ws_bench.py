"""The weight-stationary 128 -> 128 3x3 kernel (csrc/conv_ws.hip, islam_conv_ws_mode 2) against the tile kernel (mode 0) behind the same
entry point: bit-equality of the outputs, agreement of the BatchNorm partial sums, and time per call at the stereo net's shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import ops
from islam_amd._lib import lib
dev = torch.device('cuda:0')
CL = torch.channels_last


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def run(mode, x, wp, aff, stats):
    lib().islam_conv_ws_mode(mode)
    return ops.conv_nhwc(x, wp, 128, 3, in_affine=aff, stats=stats)


g = torch.Generator(device=dev).manual_seed(0)
for (B, H, W) in [(1, 4, 32), (2, 8, 64), (3, 12, 96), (5, 20, 160), (2, 112, 160), (16, 112, 160)]:
    x = torch.randn(B, 128, H, W, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=CL)
    w = (torch.randn(128, 128, 3, 3, device=dev, generator=g) / (128 * 9) ** 0.5).to(torch.bfloat16)
    wp = ops.pack_conv_nhwc_weight(w)
    aff = torch.cat([0.5 + torch.rand(128, device=dev, generator=g), 0.3 * torch.randn(128, device=dev, generator=g)]).float()
    for a in (None, aff):
        y0, f0 = run(0, x, wp, a, True)
        y2, f2 = run(2, x, wp, a, True)
        torch.cuda.synchronize()
        s0, s2 = f0.view(256, 2, 128).double().sum(0), f2.view(256, 2, 128).double().sum(0)
        print('B=%d %dx%d affine=%d: outputs equal %s (max diff %.3g), stats rel diff %.2e' % (
            B, H, W, a is not None, torch.equal(y0, y2), float((y0.float() - y2.float()).abs().max()),
            float(((s0 - s2).abs() / s0.abs().clamp_min(1e-3)).max())))
B, H, W = 16, 112, 160
fl = 2.0 * B * H * W * 128 * 128 * 9
for a in (None, aff):
    for stats in (False, True):
        t = {m: timeit(lambda: run(m, x, wp, a, stats)) for m in (0, 2, 0, 2)}
        print('affine=%d stats=%d: tile kernel %.1f us (%.0f TF/s) | weight-stationary %.1f us (%.0f TF/s)' % (
            a is not None, stats, t[0], fl / t[0] * 1e-6, t[2], fl / t[2] * 1e-6))
lib().islam_conv_ws_mode(1)
