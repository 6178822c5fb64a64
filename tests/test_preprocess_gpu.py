"""GPU test of the device-side preprocessing (islam_amd/preprocess.py; reference Datasets/utils.py:49-256,376-381 on DataLoader
workers with OpenCV) against oracle/preprocess.py: the uint8 resize is bit-exact (OpenCV's fixed-point arithmetic on both
sides), the float paths agree to float32 rounding."""
import numpy as np
import pytest
import torch

from oracle import preprocess as opre

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('H,W,calib,dtype', [(375, 1242, (718.856, 718.856, 607.1928, 185.2157), 'kitti'),
                                              (480, 752, (458.6, 457.3, 367.2, 248.4), 'euroc'), (480, 640, (320.0, 320.0, 320.0, 240.0), 'tartanair'),
                                              (300, 500, (250.0, 260.0, 251.0, 149.0), 'kitti')])
def test_make_sample_matches_the_opencv_restatement(cuda, H, W, calib, dtype):
    from islam_amd import preprocess
    g = torch.Generator().manual_seed(H + W)
    B = 2
    imgs = [torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8) for _ in range(3)]
    ext = torch.tensor([[0.54, 0, 0, 0, 0, 0, 1.0]]).repeat(B, 1)
    s = preprocess.make_sample(*[t.to(cuda) for t in imgs], torch.tensor([calib]).repeat(B, 1), ext, [dtype] * B)
    assert s['img0'].is_cuda and s['img0'].shape == (B, 3, 448, 640) and s['intrinsic'].shape == (B, 2, 112, 160)
    for b in range(B):
        ref = opre.make_sample(*[t[b].numpy() for t in imgs], calib)
        for k in ('img0', 'img1', 'img0_r'):
            got = s[k][b].cpu().numpy()
            np.testing.assert_array_equal(np.rint(got * 255).astype(np.uint8), np.rint(ref[k] * 255).astype(np.uint8))   # integer resize: bit-exact
            np.testing.assert_allclose(got, ref[k], rtol=3e-7, atol=0)                        # (/255: the device's division is 1 ulp off)
            np.testing.assert_allclose(s[k + '_norm'][b].cpu().numpy(), ref[k + '_norm'], rtol=0, atol=2e-6)
        np.testing.assert_allclose(s['intrinsic'][b].cpu().numpy(), ref['intrinsic'], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(s['intrinsic_calib'][b].numpy(), ref['intrinsic_calib'], rtol=1e-6)
    # the sample feeds TartanVO unchanged (SURVEY section 8b sample-dict contract)
    assert set(s) >= {'img0', 'img1', 'img0_r', 'img0_norm', 'img0_r_norm', 'intrinsic', 'intrinsic_calib', 'extrinsic', 'datatype'}
