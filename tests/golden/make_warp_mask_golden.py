"""Generates tests/golden/nets_warp_masks.npz by IMPORTING THE REFERENCE's PWC-Net from /root/reference (build container only):
the validity masks of the four warps of PWCDCNet.forward (Network/PWC/PWCNet.py:195-206: grid_sample(ones) >= 0.9999, pyramid
levels 5, 4, 3, 2) on the inputs of the two fixtures that sit behind a warp -- nets_pwc.npz (128x192) and nets_vonet.npz (448x640).

  python tests/golden/make_warp_mask_golden.py

tests/test_golden_gpu.py::_within uses them: the warp mask is a DISCONTINUITY of the reference network, so a reduced-precision path may
exceed its max-error bound only in the neighbourhood of a pixel whose mask decision differs from the reference's.
Same stubs as make_net_golden.py (cupy / cv2 absent; FunctionCorrelation / warp on oracle/corr81.c); the existing nets_*.npz are not
touched.  The flows are recomputed on the way and checked against the committed fixtures, so the masks belong to exactly those runs."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
for name in ('cupy', 'cv2', 'pypose'):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules['cupy'].memoize = lambda **kw: (lambda f: f)

from oracle import cwrap  # noqa: E402
from tests.golden.netfill import fill_state_dict, make_input, tame_vonet, vonet_sample  # noqa: E402

MASKS = []


def oracle_corr(tenFirst, tenSecond):
    return torch.from_numpy(cwrap.corr81_fwd(tenFirst.detach().numpy(), tenSecond.detach().numpy()))


def oracle_warp(self, x, flo):
    f = flo.detach().numpy()
    ones = np.ones((x.shape[0], 1, x.shape[2], x.shape[3]), np.float32)
    MASKS.append(cwrap.warp(ones, f)[:, 0] > 0)            # warp(ones) = grid_sample(ones) * mask: non-zero exactly where mask = 1
    return torch.from_numpy(cwrap.warp(x.detach().numpy(), f))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from Network.PWC import PWCNet as ref_pwc
    ref_pwc.FunctionCorrelation = oracle_corr
    ref_pwc.PWCDCNet.warp = oracle_warp
    out = {}
    with torch.no_grad():
        net = fill_state_dict(ref_pwc.PWCDCNet(uncertainty=False))
        del MASKS[:]
        flows, _ = net(make_input('pwc'))
        want = np.load(os.path.join(HERE, 'nets_pwc.npz'))
        np.testing.assert_allclose(flows[0].numpy(), want['flow0'], rtol=0, atol=1e-5 * float(np.abs(want['flow0']).max()))
        assert len(MASKS) == 4
        for lvl, m in zip((5, 4, 3, 2), MASKS):
            out['pwc_level%d' % lvl] = np.packbits(m, axis=None)
            out['pwc_level%d_shape' % lvl] = np.array(m.shape)
        from Network.VONet import VONet as RefVONet
        net = tame_vonet(fill_state_dict(RefVONet(fix_parts=('flow', 'stereo'))))
        net.train()
        s = vonet_sample()
        del MASKS[:]
        flows, _ = net.flowNet(torch.cat([s['img0'], s['img1']], dim=1))
        want = np.load(os.path.join(HERE, 'nets_vonet.npz'))
        np.testing.assert_allclose(flows[0].numpy(), want['flow'], rtol=0, atol=1e-5 * float(np.abs(want['flow']).max()))
        assert len(MASKS) == 4
        for lvl, m in zip((5, 4, 3, 2), MASKS):
            out['vonet_level%d' % lvl] = np.packbits(m, axis=None)
            out['vonet_level%d_shape' % lvl] = np.array(m.shape)
    np.savez_compressed(os.path.join(HERE, 'nets_warp_masks.npz'), **out)
    print({k: (v.tolist() if k.endswith('shape') else int(np.unpackbits(v).sum())) for k, v in out.items()})


if __name__ == '__main__':
    main()
