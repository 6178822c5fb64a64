"""Do the BatchNorm partial sums of the weight-stationary convolution agree with the sums over its stored output, call after call?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from islam_amd import ops
from islam_amd._lib import lib
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for (B, H, W) in [(2, 24, 40), (1, 33, 47), (3, 7, 5), (16, 112, 160), (2, 24, 40)]:
    x = torch.randn(B, 128, H, W, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 128, 3, 3, device=dev, generator=g) / (128 * 9) ** 0.5).to(torch.bfloat16)
    wp = ops.pack_conv_nhwc_weight(w)
    for mode in (0, 2):
        lib().islam_conv_ws_mode(mode)
        bad = 0
        for it in range(20):
            y, f = ops.conv_nhwc(x, wp, 128, 3, stats=True)
            s = f.view(256, 2, 128).double().sum(0)
            want = torch.stack([y.double().sum((0, 2, 3)), (y.double() ** 2).sum((0, 2, 3))])
            err = (s - want).abs() / want.abs().clamp_min(1.0)
            if float(err.max()) > 1e-4:
                bad += 1
                if bad <= 2:
                    i = int(err.argmax())
                    print('  mode %d call %d: moment %d channel %d: partial sums %.6f, output %.6f' % (mode, it, i // 128, i % 128, float(s.view(-1)[i]), float(want.view(-1)[i])))
        print('B=%d %dx%d mode %d: %d of 20 calls off' % (B, H, W, mode, bad))
lib().islam_conv_ws_mode(1)
