"""Prints the error statistics of every reduced-precision network path against the reference-generated golden vectors
(max / rms / mean signed error, relative to the reference's max / rms): the numbers the tolerances of tests/test_golden_gpu.py
are set from.  GPU box:  python scripts/calib/golden_errors.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from islam_amd import nets  # noqa: E402
from tests.golden.netfill import fill_state_dict, make_input, tame_vonet, vonet_sample  # noqa: E402

G = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden')
cuda = torch.device('cuda:0')


def stats(name, got, ref):
    got = got.detach().float().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, np.float64)
    d = got - ref
    print('%-34s max %.3e  rms %.3e  bias %.3e   (|ref| max %.3g rms %.3g)' % (
        name, np.abs(d).max() / np.abs(ref).max(), np.sqrt((d * d).mean()) / np.sqrt((ref * ref).mean()),
        d.mean() / np.sqrt((ref * ref).mean()), np.abs(ref).max(), np.sqrt((ref * ref).mean())))


ref = np.load(os.path.join(G, 'nets_pwc.npz'))
net = fill_state_dict(nets.PWCDCNet()).to(cuda).eval()
with torch.no_grad():
    f32, _ = net(make_input('pwc').to(cuda))
    mf, _ = net.forward_mfma(make_input('pwc').to(cuda))
for i in range(5):
    stats('pwc fp32 flow%d' % i, f32[i], ref['flow%d' % i])
    stats('pwc mfma flow%d' % i, mf[i], ref['flow%d' % i])

ref = np.load(os.path.join(G, 'nets_stereo.npz'))
for dt in (None, torch.bfloat16, torch.float16):
    vn = nets.VONet(fix_parts=('flow', 'stereo'))
    fill_state_dict(vn.stereoNet)
    vn = vn.to(cuda).train()
    vn.set_frozen_dtype(dt)
    with torch.no_grad():
        out = vn._run_frozen('stereo', vn.stereoNet, vn.frozen_dtype, make_input('stereo').to(cuda))[0]
    stats('stereo %s disp' % dt, out, ref['disp'])
    stats('stereo %s running_mean' % dt, vn.stereoNet.state_dict()['feature_extraction.firstconv.0.1.running_mean'], ref['running_mean_after'])

ref = np.load(os.path.join(G, 'nets_vonet.npz'))
s = vonet_sample()
args = [s[k].to(cuda) for k in ('img0', 'img1', 'img0_norm', 'img0_r_norm', 'intrinsic')]
for fd, pd in ((None, None), (torch.bfloat16, torch.bfloat16), (torch.bfloat16, None), (None, torch.bfloat16)):
    vn = tame_vonet(fill_state_dict(nets.VONet(fix_parts=('flow', 'stereo')))).to(cuda).train()
    vn.set_frozen_dtype(fd, pd)
    with torch.no_grad():
        flow, disp, pose = vn(*args)
    for k, v in (('flow', flow), ('disp', disp), ('pose', pose)):
        stats('vonet stereo=%s flow=%s %s' % (fd, pd, k), v, ref[k])
