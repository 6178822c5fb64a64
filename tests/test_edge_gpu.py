"""GPU parity of the edge-mask kernel (islam_edge_mask, reference TartanVO.py:145-155) against oracle/canny.py, the numpy
restatement of OpenCV 4.7's resize / Canny / dilate.  Integer work: bit-exact."""
import numpy as np
import pytest
import torch

from islam_amd import synthetic
from oracle import canny
from tests.helpers import edge_test_image

pytestmark = pytest.mark.gpu


def _check(img, cuda, downscale=True):
    from islam_amd import edges
    got = edges.edge_mask(img.to(cuda), downscale=downscale)
    want = canny.edge_mask(img.numpy(), downscale=downscale)
    assert got.dtype == torch.bool and tuple(got.shape) == want.shape
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    return want


@pytest.mark.parametrize('amp,cells', [(0.25, 16), (0.5, 32), (1.0, 64), (0.15, 64)])
def test_edge_mask_matches_oracle_on_textured_images(cuda, amp, cells):
    want = _check(edge_test_image(3, B=3, amp=amp, cells=cells), cuda)
    if amp >= 0.5:
        assert 0.2 < want.mean() < 0.97                # a mask that is neither empty nor full


def test_edge_mask_on_the_synthetic_stereo_batches(cuda):
    for seed in (106, 1234):
        _check(synthetic.stereo_batch(8, seed=seed)['img0'], cuda)


def test_edge_mask_step_weak_and_flat_cases(cuda):
    H, W = 64, 96
    step = torch.full((1, 3, H, W), 0.2)
    step[..., 48:] = 0.8
    flat = torch.full((1, 3, H, W), 0.5)
    # weak edge (|Sobel| between the thresholds) alone, and joined to a strong one (hysteresis carries it)
    weak = torch.full((1, 3, H, W), 80 / 255 + 1e-3)
    weak[..., 48:] = 100 / 255 + 1e-3
    joined = weak.clone()
    joined[..., 40:, 48:] = 200 / 255 + 1e-3
    for img in (step, flat, weak, joined):
        for downscale in (True, False):
            _check(img, cuda, downscale)
    assert not canny.edge_mask(weak.numpy(), downscale=False).any() and canny.edge_mask(joined.numpy(), downscale=False)[0, :30].any()
    # one channel carries the edge, first-maximum channel selection on ties
    one = torch.full((2, 3, H, W), 0.3)
    one[0, 1, :, 30:] = 0.9
    one[1, :, 20:, :] = 0.7
    _check(one, cuda)


def test_edge_mask_hysteresis_follows_long_weak_chains(cuda):
    """A serpentine of weak pixels hanging off a single strong pixel: the in-kernel relaxation must reach the fixed point the
    stack-based flood fill reaches, however long the chain."""
    H, W = 112, 160
    img = torch.full((1, 3, H, W), 60 / 255 + 1e-3)
    for k, y in enumerate(range(6, H - 6, 12)):          # horizontal bars joined alternately left / right: one long weak edge
        img[..., y:y + 6, 8:W - 8] = 80 / 255 + 1e-3
        x = W - 14 if k % 2 == 0 else 8
        img[..., y:y + 18, x:x + 6] = 80 / 255 + 1e-3
    img[..., 6:9, 8:11] = 250 / 255                      # the only strong gradients
    want = _check(img, cuda, downscale=False)
    assert want.mean() > 0.3
    ragged = edge_test_image(5, B=1, H=100, W=52, amp=1.0, cells=8, boxes=3)      # not a multiple of 64; small
    _check(ragged, cuda)
    _check(ragged[:, :, :37, :41].contiguous(), cuda, downscale=False)


def test_edge_mask_argument_checks(cuda):
    from islam_amd import _lib, edges
    with pytest.raises(_lib.IslamHipError):
        edges.edge_mask(torch.zeros(1, 3, 448, 642, device=cuda))                 # not a multiple of 4
    from islam_amd import ops
    with pytest.raises(_lib.IslamHipError):
        ops.edge_mask(torch.zeros(1, 3, 1024, 1024, device=cuda))                 # 256x256 does not fit the LDS of one CU: the kernel refuses ...
    assert edges.edge_mask(torch.zeros(1, 3, 1024, 1024, device=cuda)).shape == (1, 256, 256)      # ... and edges.edge_mask takes the tensor-op path
    with pytest.raises(RuntimeError):
        edges.edge_mask(torch.zeros(1, 3, 64, 64))                                # CPU tensor: no fallback
    assert edges.edge_mask(torch.zeros(0, 3, 64, 64, device=cuda)).shape == (0, 16, 16)


def test_images_too_large_for_one_cu_take_the_tensor_op_path(cuda):
    """More pixels than fit the LDS of one CU: same integer pipeline as device tensor ops, still bit-exact."""
    from islam_amd._lib import lib
    img = edge_test_image(5, B=1, H=192, W=256, amp=0.5, cells=32, boxes=3)
    assert 192 * 256 > lib().islam_edge_mask_max_pixels()
    _check(img, cuda, downscale=False)
    big = edge_test_image(6, B=1, H=704, W=704, amp=0.5, cells=64, boxes=3)
    assert (704 // 4) ** 2 > lib().islam_edge_mask_max_pixels()
    _check(big, cuda)
