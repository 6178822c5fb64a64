"""islam_conv3x3_mfma vs MIOpen (torch fp32 NCHW, what the flow net uses today) on PWC-Net decoder shapes at B=8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from islam_amd import ops
dev = torch.device('cuda:0')
B = 8
def timeit(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
shapes = [(6, 448, 640, 16, 2, 1), (16, 224, 320, 16, 1, 1), (16, 224, 320, 32, 2, 1), (32, 112, 160, 32, 1, 1),
          (117, 112, 160, 128, 1, 1), (245, 112, 160, 128, 1, 1), (373, 112, 160, 96, 1, 1), (469, 112, 160, 64, 1, 1),
          (533, 112, 160, 32, 1, 1), (565, 112, 160, 128, 1, 1), (128, 112, 160, 128, 1, 2), (128, 112, 160, 128, 1, 4),
          (128, 112, 160, 96, 1, 8), (181, 56, 80, 128, 1, 1), (565, 56, 80, 128, 1, 1), (32, 112, 160, 64, 2, 1),
          (64, 56, 80, 96, 2, 1), (96, 28, 40, 128, 2, 1), (64, 56, 80, 64, 1, 1), (96, 28, 40, 96, 1, 1), (529, 7, 10, 32, 1, 1)]
tot_h = tot_m = 0
for Cin, H, W, Cout, S, D in shapes:
    x = torch.randn(B, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
    b = torch.randn(Cout, device=dev)
    pk = ops.pack_conv3x3_weight(w)
    th = timeit(lambda: ops.conv3x3_mfma(x, pk, b, Cout, stride=S, dilation=D))
    tm = timeit(lambda: F.leaky_relu(F.conv2d(x, w, b, stride=S, padding=D, dilation=D), 0.1))
    fl = 2.0 * B * ((H - 1) // S + 1) * ((W - 1) // S + 1) * Cout * Cin * 9
    tot_h += th; tot_m += tm
    print('Cin=%3d %3dx%3d Cout=%3d s%d d%d: hip %7.1f us (%5.0f TF/s)   miopen fp32+lrelu %7.1f us (%5.0f TF/s)' % (
        Cin, H, W, Cout, S, D, th * 1e6, fl / th / 1e12, tm * 1e6, fl / tm / 1e12), flush=True)
print('sum: hip %.2f ms, miopen %.2f ms' % (tot_h * 1e3, tot_m * 1e3))
