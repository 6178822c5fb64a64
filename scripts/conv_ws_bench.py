"""The persistent kernels (csrc/conv_ws.hip: weight-stationary 128 -> 128; csrc/conv_ws32.hip: 32 -> 32; islam_conv_ws_mode 2) against the
tile kernel (mode 0) behind the same entry point: bit-equality of the outputs, agreement of the BatchNorm partial sums, and time per
call at the stereo net's shapes."""
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


wp_cout = {}


def run(mode, x, wp, aff, stats):
    lib().islam_conv_ws_mode(mode)
    return ops.conv_nhwc(x, wp, wp_cout[id(wp)], 3, in_affine=aff, stats=stats)


for C, CO, SHAPES in ((128, 128, [(1, 4, 32), (2, 8, 64), (3, 12, 96), (5, 20, 160), (2, 112, 160), (16, 112, 160)]),
                      (64, 128, [(1, 4, 32), (2, 8, 64), (3, 12, 96), (5, 20, 160), (2, 112, 160), (16, 112, 160)]),
                      (32, 32, [(1, 16, 32), (2, 32, 64), (3, 48, 96), (2, 224, 320), (16, 224, 320)])):
    g = torch.Generator(device=dev).manual_seed(0)
    for (B, H, W) in SHAPES:
        x = torch.randn(B, C, H, W, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=CL)
        w = (torch.randn(CO, C, 3, 3, device=dev, generator=g) / (C * 9) ** 0.5).to(torch.bfloat16)
        wp = ops.pack_conv_nhwc_weight(w)
        wp_cout[id(wp)] = CO
        aff = torch.cat([0.5 + torch.rand(C, device=dev, generator=g), 0.3 * torch.randn(C, device=dev, generator=g)]).float()
        for a in (None, aff):
            y0, f0 = run(0, x, wp, a, True)
            y2, f2 = run(2, x, wp, a, True)
            torch.cuda.synchronize()
            s0, s2 = f0.view(256, 2, CO).double().sum(0), f2.view(256, 2, CO).double().sum(0)
            print('%d->%d B=%d %dx%d affine=%d: outputs equal %s (max diff %.3g), stats rel diff %.2e' % (
                C, CO, B, H, W, a is not None, torch.equal(y0, y2), float((y0.float() - y2.float()).abs().max()),
                float(((s0 - s2).abs() / s0.abs().clamp_min(1e-3)).max())))
    fl = 2.0 * B * H * W * C * CO * 9
    for a in (None, aff):
        for stats in (False, True):
            t = {m: timeit(lambda: run(m, x, wp, a, stats)) for m in (0, 2, 0, 2)}
            print('%d->%d affine=%d stats=%d: tile kernel %.1f us (%.0f TF/s) | persistent %.1f us (%.0f TF/s)' % (
                C, CO, a is not None, stats, t[0], fl / t[0] * 1e-6, t[2], fl / t[2] * 1e-6))

lib().islam_conv_ws_mode(1)
