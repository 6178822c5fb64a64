"""Under conv_nhwc load on a side stream: launch the debug twin of the scale kernel (scripts/probes/scale_victim.hip) and, when its mask is
wrong, print what the wrong lanes READ."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from islam_amd import ops
dev = torch.device('cuda:0')
variant = int(os.environ.get('VARIANT', '0'))
iters = int(os.environ.get('ITERS', '150'))
lib = ctypes.CDLL(os.path.join(ROOT, 'islam_amd', 'lib', 'libislam_probe_scale.so'))
B, H, W = 8, 112, 160
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
cl = lambda t: t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
x = cl(rn(16, 128, 112, 160)); w = ops.pack_conv_nhwc_weight(rn(128, 128, 3, 3) / 30)
side = torch.cuda.Stream(dev)
disp = torch.full((B, 1, H, W), 10.0, device=dev)
flow = rn(B, 2, H, W)
th = torch.full((B,), 5.0, device=dev)
u = (torch.rand(B, H, W, device=dev, generator=g) > 0.5).to(torch.uint8)
mask = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
seen = torch.empty(B, H, W, 4, device=dev)
partial = torch.empty(B * 16 * 18, dtype=torch.float64, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def launch():
    rc = lib.victim_launch(P(disp), P(flow), P(u), P(th), P(mask), P(seen), P(partial), B, H, W, variant, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
launch(); torch.cuda.synchronize()
want = mask.clone()
truth = torch.stack([flow[:, 0], flow[:, 1], u.float(), disp[:, 0]], -1)
bad = shown = 0
for it in range(iters):
    with torch.cuda.stream(side):
        for _ in range(20):
            ops.conv_nhwc(x, w, 128, 3)
    for k in range(4):
        launch()
        m, s = mask.clone(), seen.clone()
        torch.cuda.synchronize()
        if not torch.equal(m, want):
            bad += 1
            if shown < 6:
                shown += 1
                idx = (m != want).nonzero()
                b, y, x0 = idx[0].tolist()
                lin = y * W + x0
                sd = (s != truth)
                print('launch %d.%d: %d wrong mask px, first (b=%d,y=%d,x=%d) lane %d; seen != truth at %d px: per field %s' % (
                    it, k, len(idx), b, y, x0, lin % 64, int(sd.any(-1).sum()), sd.sum((0, 1, 2)).tolist()))
                q = sd.any(-1).nonzero()
                for bb, yy, xx in q[:4].tolist():
                    print('     (b=%d,y=%d,x=%d) lane %2d seen %s truth %s' % (bb, yy, xx, (yy * W + xx) % 64, [round(v, 4) for v in s[bb, yy, xx].tolist()], [round(v, 4) for v in truth[bb, yy, xx].tolist()]))
print('variant %d: %d of %d launches with a wrong mask' % (variant, bad, 4 * iters))
