"""Times islam_conv_nhwc_bf16 of a given build (scripts/conv_probe.sh) on the stereo net's shapes."""
import ctypes, sys
import torch
L = ctypes.CDLL(sys.argv[1])
L.islam_conv_nhwc_bf16.restype = ctypes.c_int
L.islam_conv_nhwc_bf16.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
L.islam_conv_nhwc_packed_elems.restype = ctypes.c_size_t
dev = torch.device('cuda:0')
for (B, Cin, H, W, Cout, k) in [(16, 32, 224, 320, 32, 3), (16, 64, 112, 160, 64, 3), (16, 128, 112, 160, 128, 3), (16, 64, 112, 160, 128, 1)]:
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    wp = torch.randn(L.islam_conv_nhwc_packed_elems(Cin, Cout, k), device=dev).to(torch.bfloat16)
    y = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: L.islam_conv_nhwc_bf16(x.data_ptr(), wp.data_ptr(), None, None, None, y.data_ptr(), None, B, Cin, H, W, Cout, k, 0, s)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e3
    print('%4d->%4d k%d %dx%d: %7.1f us (%.0f TF/s if it were the whole convolution)' % (Cin, Cout, k, H, W, t, 2.0 * B * H * W * Cin * Cout * k * k / t * 1e-6))
# the flow net's kernel (fp32 NCHW activations, bf16 operands): level-2 decoder shapes at B=8
L.islam_conv3x3_mfma.restype = ctypes.c_int
L.islam_conv3x3_mfma.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 11 + [ctypes.c_float, ctypes.c_void_p]
L.islam_conv3x3_packed_elems.restype = ctypes.c_size_t
for (B, Cin, H, W, Cout) in [(8, 117, 112, 160, 128), (8, 245, 112, 160, 128), (8, 565, 112, 160, 128), (8, 469, 112, 160, 64), (8, 181, 28, 40, 128)]:
    x = torch.randn(B, Cin, H, W, device=dev)
    wp = torch.randn(L.islam_conv3x3_packed_elems(Cin, Cout), device=dev).to(torch.bfloat16)
    bias = torch.randn(Cout, device=dev)
    y = torch.empty(B, Cout, H, W, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: L.islam_conv3x3_mfma(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr(), B, Cin, H, W, Cout, 1, 1, 0, Cin, 0, Cout, 0.1, s)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e3
    print('flow %4d->%4d %dx%d: %7.1f us (%.0f TF/s if it were the whole convolution)' % (Cin, Cout, H, W, t, 2.0 * B * H * W * Cin * Cout * 9 / t * 1e-6))
