"""Packed-FP32 VALU instructions beside a matrix-core kernel on another stream (scripts/probes/pk_victim.hip)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from islam_amd import ops
dev = torch.device('cuda:0')
lib = ctypes.CDLL(os.path.join(ROOT, 'islam_amd', 'lib', 'libislam_probe_pk.so'))
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(16, 128, 112, 160, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = ops.pack_conv_nhwc_weight(torch.randn(128, 128, 3, 3, device=dev, generator=g) / 30)
xf = torch.randn(8, 64, 56, 80, device=dev, generator=g); wf = ops.pack_conv3x3_weight(torch.randn(96, 64, 3, 3, device=dev, generator=g) / 24); bf = torch.randn(96, device=dev, generator=g)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
loads = {'none': lambda: None, 'conv_nhwc': lambda: ops.conv_nhwc(x, w, 128, 3), 'conv3x3_mfma': lambda: ops.conv3x3_mfma(xf, wf, bf, 96, stride=2),
         'matmul': lambda: torch.mm(a, b)}
side = torch.cuda.Stream(dev)
for name, one in loads.items():
    bad = torch.zeros(40, dtype=torch.int32, device=dev)
    one(); torch.cuda.synchronize()
    for it in range(60):
        with torch.cuda.stream(side):
            for _ in range(20 if name != 'matmul' else 3):
                one()
        for k in range(4):
            rc = lib.pk_victim_launch(ctypes.c_void_p(bad.data_ptr()), 128, 2000, ctypes.c_float(1.0 + 0.01 * k), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
        torch.cuda.synchronize()
    h = bad.cpu().tolist()
    print('load %-13s: wrong iterations per form, by lane quarter [0-15, 16-31, 32-47, 48-63]: %s' % (name, {k: h[4 * k:4 * k + 4] for k in range(10)}))
