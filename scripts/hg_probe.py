"""Phase timestamps of ONE workgroup of hg_residual_kernel (build with -DISLAM_HG_PROBE=1: scripts/hg_probe.sh)."""
import ctypes, sys
import torch
L = ctypes.CDLL(sys.argv[1])
L.islam_hg_residual_nhwc_bf16.restype = ctypes.c_int
L.islam_hg_residual_nhwc_bf16.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
L.islam_hg_residual_packed_elems.restype = ctypes.c_size_t
L.islam_hg_probe_read.argtypes = [ctypes.c_void_p]
dev = torch.device('cuda:0')
NAMES = ['entry', 'patch staged', 'barrier 1', 'phase 1 MFMAs', 't1 written', 'barrier 2', 'phase 2 MFMAs', 't2 written', 'barrier 3',
         'phase 3 + staging', 'barrier 4', 'stored']
for (B, Cin, Cout, H, W) in [(8, 256, 256, 14, 20), (8, 256, 256, 28, 40), (8, 192, 192, 56, 80), (8, 128, 128, 112, 160)]:
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    wp = (torch.randn(L.islam_hg_residual_packed_elems(Cin, Cout), device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.zeros(2 * Cout, device=dev)
    res = x if Cin == Cout else torch.randn(B, H, W, Cout, device=dev).to(torch.bfloat16)
    y = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert L.islam_hg_residual_nhwc_bf16(x.data_ptr(), res.data_ptr(), y.data_ptr(), wp.data_ptr(), bias.data_ptr(), B, Cin, H, W, Cout, s) == 0
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    assert L.islam_hg_probe_read(buf) == 0
    b = list(buf)
    t0 = min(v for v in b[0::16] if v)
    print('%d->%d %dx%d (us since the first wave entered):' % (Cin, Cout, H, W))
    for w in range(4):
        if b[16 * w]:
            print('  wave %d: ' % w + '  '.join('%s %.2f' % (NAMES[i], (b[16 * w + i] - t0) / 100.0) for i in range(12)))
