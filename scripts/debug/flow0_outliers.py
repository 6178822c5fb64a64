"""Where does flow0's maximum error against the golden vector sit?  Error map of PWCDCNet.forward_mfma's full-resolution flow, the
fraction of pixels above thresholds, their positions, and the level-2 warp (zeroed by the validity mask or not) at those positions
for the default path and for ISLAM_FLOW_HEAD_MIRROR=0 run in the same process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.golden.netfill import fill_state_dict, make_input
from tests.test_golden_gpu import _g, _stats
from islam_amd import nets
print(torch.cuda.get_device_name(0))
ref = _g('pwc')
x = make_input('pwc').to('cuda')
out = {}
for mirror in (True, False):
    nets.FLOW_HEAD_MIRROR = mirror
    net = fill_state_dict(nets.PWCDCNet()).to('cuda').eval()
    warps = []
    orig = nets.warp_fn
    nets.warp_fn = lambda f, fl, sc: (warps.append(orig(f, fl, sc)), warps[-1])[1]
    with torch.no_grad():
        flows, _ = net.forward_mfma(x)
    nets.warp_fn = orig
    out[mirror] = (flows[0].float().cpu().numpy(), [w.float().cpu().numpy() for w in warps])
r0 = np.asarray(ref['flow0'], np.float64)
mx = np.abs(r0).max()
for mirror in (True, False):
    f0, warps = out[mirror]
    e = np.abs(f0 - r0).max(1)[0] / mx                       # (H, W): worst channel
    print('mirror head %s: max %.3e; pixels above 2e-2: %d, above 5e-2: %d of %d; 99.9th percentile %.3e'
          % (mirror, e.max(), (e > 2e-2).sum(), (e > 5e-2).sum(), e.size, np.quantile(e, 0.999)))
    ys, xs = np.where(e > 5e-2)
    print('   positions (y, x) of errors above 5e-2:', list(zip(ys.tolist(), xs.tolist()))[:20])
wa, wb = out[True][1][-1], out[False][1][-1]               # level-2 warp (last of the four)
za, zb = (np.abs(wa).sum(1) == 0)[0], (np.abs(wb).sum(1) == 0)[0]
print('level-2 warp %s: masked-out pixels %d (mirror head) / %d (fp32 head); pixels masked in one and not the other: %d at %s'
      % (wa.shape, za.sum(), zb.sum(), (za != zb).sum(), list(zip(*np.where(za != zb)))[:12]))
