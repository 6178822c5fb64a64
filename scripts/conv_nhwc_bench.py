"""islam_conv_nhwc_bf16 against MIOpen (torch bf16 channels_last conv, kernel search on) on the stereo feature extractor's
shapes at B=8 (16 images): time per call, TFLOP/s, effective HBM GB/s of the algorithmic traffic (read x once, write y once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from islam_amd import ops
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = True
CL = torch.channels_last
shapes = [(16, 32, 224, 320, 32, 3), (16, 64, 112, 160, 64, 3), (16, 64, 112, 160, 128, 3), (16, 128, 112, 160, 128, 3),
          (16, 352, 224, 320, 128, 3), (16, 128, 224, 320, 64, 1), (16, 64, 112, 160, 128, 1)]


def timeit(fn, n=20):
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


tot_m = tot_h = 0.0
for (B, Cin, H, W, Cout, k) in shapes:
    x = torch.randn(B, Cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=CL)
    w = (torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5).to(torch.bfloat16).contiguous(memory_format=CL)
    wp = ops.pack_conv_nhwc_weight(w)
    t_m = timeit(lambda: F.conv2d(x, w, None, 1, k // 2))
    t_h = timeit(lambda: ops.conv_nhwc(x, wp, Cout, k))
    t_s = timeit(lambda: ops.conv_nhwc(x, wp, Cout, k, stats=True))
    fl = 2.0 * B * H * W * Cout * Cin * k * k
    by = 2.0 * B * H * W * (Cin + Cout)
    tot_m += t_m
    tot_h += t_s
    print('%4d->%4d k%d %3dx%3d  MIOpen %7.1f us | HIP %7.1f us (%5.1f TF/s, %4.0f GB/s) | HIP+stats %7.1f us' % (
        Cin, Cout, k, H, W, t_m, t_h, fl / t_h * 1e-6, by / t_h * 1e-3, t_s))
print('sum: MIOpen %.1f us, HIP+stats %.1f us' % (tot_m, tot_h))
