"""CPU tests: the re-implemented network definitions reproduce the REFERENCE modules' outputs.

Golden vectors were produced by importing the reference's own modules (tests/golden/make_net_golden.py);
weights and inputs are regenerated from seeds (tests/golden/netfill.py).  Float32 tolerances: the same conv
stacks evaluated in the same order -- differences come only from oneDNN/MKL kernel selection."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cwrap
from tests.golden.netfill import fill_state_dict, make_input

G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def nets():
    from islam_amd import nets as N
    return N


def test_state_dict_keys_match_reference(nets):
    keys = json.load(open(os.path.join(G, 'nets_keys.json')))
    vonet = nets.VONet(fix_parts=('flow', 'stereo'))
    sd = vonet.state_dict()
    assert len(sd) == 765
    for part in ('flowNet', 'stereoNet', 'flowPoseNet'):
        mine = {k[len(part) + 1:]: list(v.shape) for k, v in sd.items() if k.startswith(part + '.')}
        assert mine == keys[part], part
    den = nets.IMUCorrector_CNN_GRU_WO_COV()
    assert {k: list(v.shape) for k, v in den.state_dict().items()} == keys['denoiser']
    # frozen parts (Network/VONet.py:20-26) and the trainable pose head
    assert not any(p.requires_grad for p in vonet.flowNet.parameters())
    assert not any(p.requires_grad for p in vonet.stereoNet.parameters())
    assert all(p.requires_grad for p in vonet.flowPoseNet.parameters())


def test_pose_net_matches_reference(nets):
    ref = np.load(os.path.join(G, 'nets_pose.npz'))
    net = fill_state_dict(nets.VOFlowRes(fix_parts=('flow', 'stereo')))
    with torch.no_grad():
        out = net(make_input('pose')).numpy()
    np.testing.assert_allclose(out, ref['pose'], rtol=1e-4, atol=1e-5)


def test_stereo_net_train_mode_bn_matches_reference(nets):
    ref = np.load(os.path.join(G, 'nets_stereo.npz'))
    net = fill_state_dict(nets.StereoNet7())
    net.train()                                       # TartanVO.py:91: BN uses batch statistics
    with torch.no_grad():
        out = net(make_input('stereo'))[0].numpy()
    scale = np.abs(ref['disp']).max()
    assert np.abs(out - ref['disp']).max() <= 2e-4 * scale
    rm = net.state_dict()['feature_extraction.firstconv.0.1.running_mean'].numpy()
    np.testing.assert_allclose(rm, ref['running_mean_after'], rtol=1e-4, atol=1e-6)   # running stats drift like the reference


def test_pwc_net_matches_reference(nets, monkeypatch):
    ref = np.load(os.path.join(G, 'nets_pwc.npz'))
    monkeypatch.setattr(nets, 'corr_fn', lambda a, b: torch.from_numpy(cwrap.corr81_fwd(a.numpy(), b.numpy())))
    monkeypatch.setattr(nets, 'warp_fn', lambda x, f, s: torch.from_numpy(cwrap.warp(x.numpy(), (f * s).numpy())))
    net = fill_state_dict(nets.PWCDCNet())
    with torch.no_grad():
        flows, _ = net(make_input('pwc'))
    for i, f in enumerate(flows):
        r = ref['flow%d' % i]
        assert np.abs(f.numpy() - r).max() <= 2e-4 * max(np.abs(r).max(), 1e-3), i


def test_imu_denoiser_matches_reference(nets):
    ref = np.load(os.path.join(G, 'nets_denoise.npz'))
    net = fill_state_dict(nets.IMUCorrector_CNN_GRU_WO_COV())
    ca, cg, _, _ = net({'acc': make_input('acc'), 'gyro': make_input('gyro')}, eval=True)
    np.testing.assert_allclose(ca.numpy(), ref['cacc'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cg.numpy(), ref['cgyro'], rtol=1e-5, atol=1e-6)
    assert not ca.requires_grad                       # eval=True disables grad


def test_whole_vonet_matches_reference(nets, monkeypatch):
    """Network/VONet.py:28-39 at the production size (448x640), train mode (TartanVO.py:91)."""
    from tests.golden.netfill import tame_vonet, vonet_sample
    ref = np.load(os.path.join(G, 'nets_vonet.npz'))
    monkeypatch.setattr(nets, 'corr_fn', lambda a, b: torch.from_numpy(cwrap.corr81_fwd(a.numpy(), b.numpy())))
    monkeypatch.setattr(nets, 'warp_fn', lambda x, f, s: torch.from_numpy(cwrap.warp(x.numpy(), (f * s).numpy())))
    net = tame_vonet(fill_state_dict(nets.VONet(fix_parts=('flow', 'stereo')))).train()
    s = vonet_sample()
    with torch.no_grad():
        flow, disp, pose = net(*[s[k] for k in ('img0', 'img1', 'img0_norm', 'img0_r_norm', 'intrinsic')])
    for got, key, tol in ((flow, 'flow', 2e-4), (disp, 'disp', 2e-4), (pose, 'pose', 1e-3)):
        assert np.abs(got.numpy() - ref[key]).max() <= tol * max(np.abs(ref[key]).max(), 1e-3), key


def test_quarter_resolution_tail_computes_exactly_the_pixels_vonet_keeps(nets):
    """StereoNet7.forward(quarter=True) == forward()[..., ::4, ::4] == what Network/VONet.py:33-34 keeps of the disparity
    (F.interpolate(scale_factor=0.25, mode='nearest')): the 4x4 stride-2 deconvolution restricted to every 4th output pixel is a
    2x2 stride-2 convolution, the two 1x1 convolutions after it are pointwise."""
    torch.manual_seed(1)
    net = fill_state_dict(nets.StereoNet7()).eval()
    x = make_input('stereo')[:1]
    with torch.no_grad():
        full = net(x)[0]
        q = net(x, quarter=True)[0]
        dq = net._deconv_c11_quarter(torch.randn(2, 128, 18, 26))
        dfull = net.deconv_c11(torch.randn(2, 128, 18, 26, generator=torch.Generator().manual_seed(9)))
    kept = torch.nn.functional.interpolate(full, scale_factor=0.25, mode='nearest')
    assert q.shape == kept.shape == (1, 1, 64, 64)
    assert float((q - kept).abs().max()) <= 1e-5 * float(kept.abs().max())
    assert dq.shape == (2, 64, 9, 13) and dfull.shape == (2, 64, 36, 52)
    xin = torch.randn(2, 128, 18, 26, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        torch.testing.assert_close(net._deconv_c11_quarter(xin), net.deconv_c11(xin)[..., ::4, ::4], rtol=1e-4, atol=1e-4 * float(dfull.abs().max()))
