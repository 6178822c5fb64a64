"""islam_upsample_cat_nhwc_bf16 at the stereo net's shape (six pieces = 320 channels at 112x160x16 up to 224x320, 32-channel tail):
time per call and bit-equality against one resize per piece + a copy of the tail.  ISLAM_UPCAT_ROWS=0: the one-thread-per-group kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import ops
dev = torch.device('cuda:0')
CL = torch.channels_last
g = torch.Generator(device=dev).manual_seed(0)
B, Hi, Wi = 16, 112, 160
mk = lambda c, h, w: torch.randn(B, c, h, w, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=CL)
pieces = [mk(64, Hi, Wi), mk(128, Hi, Wi)] + [mk(32, Hi, Wi) for _ in range(4)]
tail = mk(32, 2 * Hi, 2 * Wi)
got = ops.upsample_cat(pieces, (2 * Hi, 2 * Wi), tail=tail, align_corners=True)
want = torch.cat([ops.resize_bilinear(p, [2 * Hi, 2 * Wi], align_corners=True) for p in pieces] + [tail], 1)
print('bit-equal to resize + cat:', torch.equal(got, want))
for _ in range(3):
    ops.upsample_cat(pieces, (2 * Hi, 2 * Wi), tail=tail, align_corners=True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
N = int(os.environ.get('UPCAT_ITERS', '20'))
for _ in range(N):
    ops.upsample_cat(pieces, (2 * Hi, 2 * Wi), tail=tail, align_corners=True)
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / N * 1e3
by = got.numel() * 2 + sum(p.numel() for p in pieces) * 2 + tail.numel() * 2
print('ISLAM_UPCAT_ROWS=%s: %.1f us per call, %.2f TB/s of the algorithmic %.0f MB' % (os.environ.get('ISLAM_UPCAT_ROWS', '1'), t, by / t * 1e-6, by / 1e6))

if os.environ.get('UPCAT_ITERS'):
    sys.exit(0)


def tm(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


big = torch.empty_like(got)
t_fill = tm(lambda: big.zero_())
t_copy = tm(lambda: big.copy_(got))
print('for scale: zero-fill of the %.0f MB output %.1f us (%.2f TB/s written); copy of it %.1f us (%.2f TB/s read + written)' % (
    got.numel() * 2 / 1e6, t_fill, got.numel() * 2 / t_fill * 1e-6, t_copy, 2 * got.numel() * 2 / t_copy * 1e-6))
