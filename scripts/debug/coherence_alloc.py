"""islam_scale_ls under conv_nhwc load: (a) ops.scale_ls as the product calls it (fresh torch.empty outputs per call, .bool() behind it),
(b) the same C entry point on FIXED output buffers + clone, (c) fixed buffers, results read back with .bool() like the product."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from islam_amd import ops, _lib
from islam_amd._lib import ptr, stream_ptr, check
import islam_amd._lib as L
_alt = os.environ.get('ALT')
if _alt:
    _c = ctypes.CDLL(os.path.join(ROOT, 'islam_amd', 'lib', _alt))
    _c.islam_scale_ls.argtypes = L.SIGNATURES['islam_scale_ls'][1]
    lib = lambda: _c
else:
    lib = L.lib
dev = torch.device('cuda:0')
iters = int(os.environ.get('ITERS', '150'))
B, H, W = 8, 112, 160
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
cl = lambda t: t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
x = cl(rn(16, 128, 112, 160)); w = ops.pack_conv_nhwc_weight(rn(128, 128, 3, 3) / 30)
side = torch.cuda.Stream(dev)
disp = torch.full((B, 1, H, W), 10.0, device=dev)
flow = rn(B, 2, H, W)
pose7 = torch.tensor([[0.1, 0.2, 1.0, 0, 0, 0, 1.0]] * B, device=dev)
intr4 = torch.tensor([[180.0, 180.0, 80.0, 56.0]] * B).to(dev)
baseline = torch.full((B,), 0.5).to(dev)
th = torch.full((B,), 5.0, device=dev)
u = (torch.rand(B, H, W, device=dev, generator=g) > 0.5).to(torch.uint8)
fix = dict(scale=torch.empty(B, device=dev), z=torch.empty(B, H, W, device=dev), mask=torch.empty(B, H, W, dtype=torch.uint8, device=dev),
           dmask=torch.empty(B, H, W, dtype=torch.uint8, device=dev), sums=torch.empty(B, _lib.SCALE_NSUM, dtype=torch.float64, device=dev),
           partial=torch.empty(B, _lib.SCALE_NBLK, _lib.SCALE_NSUM, dtype=torch.float64, device=dev))
def fixed(readback):
    check(lib().islam_scale_ls(ptr(disp), ptr(flow), ptr(pose7), ptr(intr4), ptr(baseline), ptr(u), ptr(th), ptr(fix['scale']), ptr(fix['z']),
                               ptr(fix['mask']), ptr(fix['dmask']), ptr(fix['sums']), ptr(fix['partial']), B, H, W, stream_ptr(dev)))
    return (fix['mask'].bool() if readback == 'bool' else fix['mask'].clone()), fix['scale'].clone()
modes = {'fixed_clone': lambda: fixed('clone')}
want = {}
for k, f in modes.items():
    m, s = f(); torch.cuda.synchronize(); want[k] = (m.clone(), s.clone())
bad = {k: [0, 0] for k in modes}
for it in range(iters):
    for k, f in modes.items():
        with torch.cuda.stream(side):
            for _ in range(20):
                ops.conv_nhwc(x, w, 128, 3)
        rs = [f() for _ in range(4)]
        torch.cuda.synchronize()
        bad[k][0] += sum(int(not torch.equal(m, want[k][0])) for m, s in rs)
        bad[k][1] += sum(int(not torch.equal(s, want[k][1])) for m, s in rs)
print(_alt, 'wrong [mask, scale] of %d launches: %s' % (4 * iters, bad))

# ---- raw dump of a failing launch: the uint8 mask / dmask bytes and z around the first wrong pixel
if os.environ.get('DUMP') == '1':
    wm, _ = want['fixed_clone']
    check(lib().islam_scale_ls(ptr(disp), ptr(flow), ptr(pose7), ptr(intr4), ptr(baseline), ptr(u), ptr(th), ptr(fix['scale']), ptr(fix['z']),
                               ptr(fix['mask']), ptr(fix['dmask']), ptr(fix['sums']), ptr(fix['partial']), B, H, W, stream_ptr(dev)))
    torch.cuda.synchronize()
    wz, wd, wp = fix['z'].clone(), fix['dmask'].clone(), fix['partial'].clone()
    shown = 0
    for it in range(400):
        with torch.cuda.stream(side):
            for _ in range(20):
                ops.conv_nhwc(x, w, 128, 3)
        for k in range(4):
            fix['mask'].fill_(0x77); fix['dmask'].fill_(0x77); fix['z'].fill_(-7.0); fix['partial'].fill_(-7.0)
            check(lib().islam_scale_ls(ptr(disp), ptr(flow), ptr(pose7), ptr(intr4), ptr(baseline), ptr(u), ptr(th), ptr(fix['scale']), ptr(fix['z']),
                                       ptr(fix['mask']), ptr(fix['dmask']), ptr(fix['sums']), ptr(fix['partial']), B, H, W, stream_ptr(dev)))
            m, z_, d_, p_ = fix['mask'].clone(), fix['z'].clone(), fix['dmask'].clone(), fix['partial'].clone()
            torch.cuda.synchronize()
            if not torch.equal(m, wm) and shown < 4:
                shown += 1
                idx = (m != wm).nonzero()
                b, y, x0 = idx[0].tolist()
                lin = y * W + x0
                lo = lin // 64 * 64
                f = lambda t: t[b].reshape(-1)[lo:lo + 64].tolist()
                print('wrong mask px %d, z px %d, dmask px %d, partial entries %d; first (b=%d, i=%d) wave chunk [%d, %d)' % (
                    len(idx), int((z_ != wz).sum()), int((d_ != wd).sum()), int((p_ != wp).sum()), b, lin, lo, lo + 64))
                print('  mask got ', ''.join('%x' % min(v, 15) for v in f(m)))
                print('  mask want', ''.join('%x' % min(v, 15) for v in f(wm)))
                print('  dmask got', ''.join('%x' % min(v, 15) for v in f(d_)))
                print('  z got/want lanes 44..63', [round(v, 2) for v in f(z_)[44:]], [round(v, 2) for v in f(wz)[44:]])
                pb = (p_ != wp).nonzero()
                print('  partial diffs (b, blk, k):', pb[:6].tolist())
