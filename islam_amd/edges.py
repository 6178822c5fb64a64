"""Edge mask of TartanVO.forward on the device (reference TartanVO.py:145-155).

The reference copies img0 to the host (27.5 MB D2H at B=8), loops over the batch with
cv2.resize(1/4) -> cv2.Canny(50, 100) -> cv2.dilate(5x5) and copies the mask back: a full device
synchronisation in the middle of the forward.  Here the whole pipeline is ONE HIP launch
(islam_edge_mask, islam_amd/csrc/edge_mask.hip: one workgroup per image, the quarter-resolution image,
gradient magnitudes and the hysteresis map in LDS, hysteresis iterated to its fixed point inside the
kernel) -- no host round trip, no synchronisation.  The integer arithmetic is OpenCV 4.7's
(restated for the tests in oracle/canny.py; opencv is not installable in the build container, so this
piece is "parity unpinned").

Images whose edge map does not fit the LDS of one CU (more than islam_edge_mask_max_pixels() pixels after the
resize, e.g. 1024x1024 sources or downscale=False at 448x640) take `_edge_mask_tensor_ops`: the same integer
pipeline as device tensor ops (one convergence check per 16 hysteresis sweeps is a host sync; this path is off
the benchmarked 448x640 configuration).
"""
import torch
import torch.nn.functional as F

from . import ops
from ._lib import lib, require_cuda

_TG22 = 13573      # round(tan(22.5 deg) * 2**15)


def _shift(x, dy, dx, fill=0):
    """x[..., i + dy, j + dx] with `fill` outside the image."""
    H, W = x.shape[-2:]
    p = F.pad(x, (1, 1, 1, 1), value=fill)
    return p[..., 1 + dy:1 + dy + H, 1 + dx:1 + dx + W]


def _edge_mask_tensor_ops(img0, downscale=True, low=50, high=100):
    """(B,3,H,W) float32 in [0,1] -> (B,h,w) bool; OpenCV 4.7's resize / Canny / dilate in integer tensor ops."""
    u8 = (img0.float() * 255.0).to(torch.uint8).to(torch.int32)              # .astype(np.uint8): truncation
    if downscale:
        assert u8.shape[-2] % 4 == 0 and u8.shape[-1] % 4 == 0, 'the exact-1/4 resize needs H, W multiples of 4'
        s = u8[..., 1::4, 1::4] + u8[..., 1::4, 2::4] + u8[..., 2::4, 1::4] + u8[..., 2::4, 2::4]
        u8 = (s + 2) >> 2
    p = F.pad(u8.float(), (1, 1, 1, 1), mode='replicate').to(torch.int32)   # Sobel 3x3, BORDER_REPLICATE (values < 2^24: exact)
    tl, tc, tr = p[..., :-2, :-2], p[..., :-2, 1:-1], p[..., :-2, 2:]
    ml, mr = p[..., 1:-1, :-2], p[..., 1:-1, 2:]
    bl, bc, br = p[..., 2:, :-2], p[..., 2:, 1:-1], p[..., 2:, 2:]
    dxs = (tr + 2 * mr + br) - (tl + 2 * ml + bl)
    dys = (bl + 2 * bc + br) - (tl + 2 * tc + tr)
    dx, dy = dxs[:, 0], dys[:, 0]
    mag = dx.abs() + dy.abs()
    for c in range(1, u8.shape[1]):                                          # the FIRST channel holding the maximum wins
        n = dxs[:, c].abs() + dys[:, c].abs()
        take = n > mag
        mag, dx, dy = torch.where(take, n, mag), torch.where(take, dxs[:, c], dx), torch.where(take, dys[:, c], dy)
    ax, ay = dx.abs().to(torch.int64), dy.abs().to(torch.int64) << 15
    tg22 = ax * _TG22
    tg67 = tg22 + (ax << 16)
    neg = (dx ^ dy) < 0                                                      # s = -1: compare with up-right / down-left
    m = mag
    c = lambda oy, ox: _shift(mag, oy, ox, 0)                                # magnitude buffer is zero outside the image
    keep_h = (m > c(0, -1)) & (m >= c(0, 1))
    keep_v = (m > c(-1, 0)) & (m >= c(1, 0))
    keep_d = torch.where(neg, (m > c(-1, 1)) & (m > c(1, -1)), (m > c(-1, -1)) & (m > c(1, 1)))
    keep = torch.where(ay < tg22, keep_h, torch.where(ay > tg67, keep_v, keep_d)) & (m > low)
    edges = keep & (m > high)
    keep_f = keep.float().unsqueeze(1)
    e = edges.float().unsqueeze(1)
    while True:                                                              # 8-connected hysteresis to its fixed point
        prev = e
        for _ in range(16):
            e = F.max_pool2d(e, 3, 1, 1) * keep_f
        if torch.equal(e, prev):
            break
    return F.max_pool2d(e, 5, 1, 2)[:, 0] > 0                                # 5x5 dilation, border never wins


def edge_mask(img0, downscale=True):
    """img0: (B,3,H,W) float in [0,1] (BGR/255 like the reference) -> (B,H/4,W/4) bool."""
    require_cuda(img0)
    H, W = img0.shape[-2:]
    n = (H // 4) * (W // 4) if downscale else H * W
    if n > lib().islam_edge_mask_max_pixels():
        return _edge_mask_tensor_ops(img0, downscale, 50, 100)
    return ops.edge_mask(img0, downscale=downscale, low=50, high=100)
