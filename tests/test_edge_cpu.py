"""The large-image edge-mask path of islam_amd/edges.py (device tensor ops; reference TartanVO.py:145-155) is plain integer
torch arithmetic, so its agreement with oracle/canny.py can be checked without a GPU.  (The public `edge_mask` refuses CPU
tensors; the product has no CPU path.)"""
import numpy as np
import pytest
import torch

from oracle import canny
from tests.helpers import edge_test_image


@pytest.mark.parametrize('downscale,H,W', [(True, 128, 192), (False, 64, 96)])
def test_tensor_op_edge_mask_is_bit_exact(downscale, H, W):
    from islam_amd.edges import _edge_mask_tensor_ops
    img = edge_test_image(11, B=2, H=H, W=W, amp=0.5, cells=32, boxes=3)
    got = _edge_mask_tensor_ops(img, downscale=downscale).numpy()
    np.testing.assert_array_equal(got, canny.edge_mask(img.numpy(), downscale=downscale))


def test_public_edge_mask_refuses_cpu_tensors():
    from islam_amd import edges
    with pytest.raises(RuntimeError):
        edges.edge_mask(torch.zeros(1, 3, 64, 64))
