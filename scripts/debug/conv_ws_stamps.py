"""Phase clocks of the weight-stationary convolution (one workgroup's wave 0): build the stamped library first,
  cd islam_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DISLAM_WS_STAMPS -c conv_ws.hip -o build/conv_ws_stamps.o &&
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libislam_probe_ws.so $(ls build/*.o | grep -v 'conv_ws\\.o') -L/opt/rocm/lib -lrccl
then ISLAM_HIP_LIB=islam_amd/lib/libislam_probe_ws.so python scripts/debug/conv_ws_stamps.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from islam_amd import ops
from islam_amd._lib import lib
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
B, H, W = 16, 112, 160
x = torch.randn(B, 128, H, W, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(128, 128, 3, 3, device=dev, generator=g) / (128 * 9) ** 0.5).to(torch.bfloat16)
wp = ops.pack_conv_nhwc_weight(w)
aff = torch.cat([0.5 + torch.rand(128, device=dev, generator=g), 0.3 * torch.randn(128, device=dev, generator=g)]).float()
L = ctypes.CDLL(os.environ['ISLAM_HIP_LIB'])
lib().islam_conv_ws_mode(2)
names = ['multiply phase + riders', 'barrier A', 'accumulators -> LDS', 'barrier B', 'last store phase (whole loop)', 'loop top']
for a in (None, aff):
    for rep in range(3):
        for _ in range(10):
            ops.conv_nhwc(x, wp, 128, 3, in_affine=a)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16)()
        L.islam_conv_ws_stamps(buf)
        n = buf[6]
        tot = sum(buf[j] for j in range(6))
        print('affine=%d: %d tiles, per tile (shader clocks): ' % (a is not None, n) + ', '.join('%s %d' % (names[j], buf[j] // max(n, 1)) for j in range(6))
              + ' | loop %.1f us wall, %.2f GHz | weights %.1f us, first tile staged %.1f us' % (buf[7] / 100.0, tot / (buf[7] * 10.0), buf[8] / 100.0, buf[9] / 100.0))
lib().islam_conv_ws_mode(1)
