"""One large PWC decoder layer through islam_conv3x3_mfma, repeated (target of PMC passes): Cin Cout H W from argv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import ops
dev = torch.device('cuda:0')
Cin, Cout, H, W = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (565, 128, 112, 160))]
B = 8
x = torch.randn(B, Cin, H, W, device=dev)
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
b = torch.randn(Cout, device=dev)
pk = ops.pack_conv3x3_weight(w)
for _ in range(10):
    y = ops.conv3x3_mfma(x, pk, b, Cout)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y = ops.conv3x3_mfma(x, pk, b, Cout)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e-3
print('Cin=%d Cout=%d %dx%d: %.1f us, %.0f TF/s' % (Cin, Cout, H, W, t * 1e6, 2.0 * B * H * W * Cout * Cin * 9 / t / 1e12))
