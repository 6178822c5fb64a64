"""Phase timestamps of ONE workgroup of conv_nhwc_kernel (build with -DISLAM_CONV_PROBE=3: scripts/conv_phase_probe.sh)."""
import ctypes, sys
import torch
L = ctypes.CDLL(sys.argv[1])
L.islam_conv_nhwc_bf16.restype = ctypes.c_int
L.islam_conv_nhwc_bf16.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
L.islam_conv_nhwc_packed_elems.restype = ctypes.c_size_t
L.islam_conv_probe_read.argtypes = [ctypes.c_void_p]
dev = torch.device('cuda:0')
for (B, Cin, H, W, Cout, k) in [(16, 352, 224, 320, 128, 3), (16, 128, 112, 160, 128, 3), (16, 64, 112, 160, 64, 3)]:
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    wp = torch.randn(L.islam_conv_nhwc_packed_elems(Cin, Cout, k), device=dev).to(torch.bfloat16)
    y = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert L.islam_conv_nhwc_bf16(x.data_ptr(), wp.data_ptr(), None, None, None, y.data_ptr(), None, B, Cin, H, W, Cout, k, 0, s) == 0
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 128)()
    assert L.islam_conv_probe_read(buf) == 0
    b = list(buf)
    us = lambda i: (b[i] - b[0]) / 100.0
    n = (Cin + 31) // 32
    print('%d->%d k%d %dx%d: set-up done %.2f us' % (Cin, Cout, k, H, W, us(1)))
    for c in range(n):
        o = 2 + 5 * c
        print('  chunk %2d: top %.2f | barrier passed %.2f | staged %.2f | barrier passed %.2f | multiplied %.2f   (stage %.2f, multiply %.2f)' % (
            c, us(o), us(o + 1), us(o + 2), us(o + 3), us(o + 4), us(o + 2) - us(o + 1), us(o + 4) - us(o + 3)))
    print('  epilogue start %.2f  end %.2f us;  shader clock over the workgroup\'s life: %.0f MHz' % (us(120), us(121), (b[125] - b[124]) / us(121)))
