"""Edge mask of TartanVO.forward on the device (reference TartanVO.py:145-155).

The reference copies img0 to the host (27.5 MB D2H at B=8), loops over the batch with
cv2.resize(1/4) -> cv2.Canny(50, 100) -> cv2.dilate(5x5) and copies the mask back.  Here the same
pipeline runs as integer tensor ops on the GPU (no host round trip).  It restates OpenCV 4.7's
algorithm (opencv is absent from the build container, so this piece is
"parity unpinned" and covered by property tests only):
  resize   : INTER_LINEAR at exactly 1/4 = mean of the centre 2x2 of every 4x4 cell, round half up
  Canny    : Sobel 3x3 (replicated border) per channel, channel with the largest |dx|+|dy| wins,
             L1 magnitude, non-maximum suppression with the tan(22.5 deg) fixed-point sector test,
             double threshold (mag > high strong, > low weak), 8-connected hysteresis
  dilate   : 5x5 ones
"""
import torch
import torch.nn.functional as F

_TG22 = 13573      # round(tan(22.5 deg) * 2**15)


def quarter_resize_u8(img_u8):
    """(B,C,H,W) uint8 -> (B,C,H/4,W/4) uint8."""
    x = img_u8.to(torch.int32)
    s = x[..., 1::4, 1::4] + x[..., 1::4, 2::4] + x[..., 2::4, 1::4] + x[..., 2::4, 2::4]
    return ((s + 2) >> 2).to(torch.uint8)


def _sobel(x):
    """x: (B,C,H,W) int32 -> dx, dy with replicated borders."""
    p = F.pad(x.float(), (1, 1, 1, 1), mode='replicate').to(torch.int32)
    tl, tc, tr = p[..., :-2, :-2], p[..., :-2, 1:-1], p[..., :-2, 2:]
    ml, mr = p[..., 1:-1, :-2], p[..., 1:-1, 2:]
    bl, bc, br = p[..., 2:, :-2], p[..., 2:, 1:-1], p[..., 2:, 2:]
    dx = (tr + 2 * mr + br) - (tl + 2 * ml + bl)
    dy = (bl + 2 * bc + br) - (tl + 2 * tc + tr)
    return dx, dy


def canny_u8(img_u8, low=50, high=100):
    """(B,C,H,W) uint8 -> (B,H,W) bool edge map."""
    dx, dy = _sobel(img_u8.to(torch.int32))
    mag_c = dx.abs() + dy.abs()
    idx = mag_c.argmax(dim=1, keepdim=True)                  # first channel with the maximal magnitude
    dx, dy = dx.gather(1, idx)[:, 0], dy.gather(1, idx)[:, 0]
    mag = mag_c.gather(1, idx)[:, 0]
    mp = F.pad(mag, (1, 1, 1, 1), value=0)
    c = lambda oy, ox: mp[:, 1 + oy:mp.shape[1] - 1 + oy, 1 + ox:mp.shape[2] - 1 + ox]
    ax, ay = dx.abs().to(torch.int64), dy.abs().to(torch.int64) << 15
    tg22 = ax * _TG22
    tg67 = tg22 + (ax << 16)
    horiz = ay < tg22
    vert = ay > tg67
    s_neg = (dx ^ dy) < 0                                    # diagonal orientation
    m = mag
    keep_h = (m > c(0, -1)) & (m >= c(0, 1))
    keep_v = (m > c(-1, 0)) & (m >= c(1, 0))
    keep_d = torch.where(s_neg, (m > c(-1, 1)) & (m > c(1, -1)), (m > c(-1, -1)) & (m > c(1, 1)))
    keep = torch.where(horiz, keep_h, torch.where(vert, keep_v, keep_d)) & (m > low)
    strong = keep & (m > high)
    edges = strong
    for _ in range(64):                                      # hysteresis: grow through weak candidates until stable
        prev = edges
        for _ in range(16):
            edges = (F.max_pool2d(edges.float().unsqueeze(1), 3, 1, 1)[:, 0] > 0) & keep
        if torch.equal(edges, prev):
            break
    return edges


def edge_mask(img0, downscale=True):
    """img0: (B,3,H,W) float in [0,1] (BGR/255 like the reference) -> (B,H/4,W/4) bool, TartanVO.py:145-155."""
    u8 = (img0 * 255).to(torch.uint8)                        # .astype(np.uint8) truncates
    if downscale:
        u8 = quarter_resize_u8(u8)
    e = canny_u8(u8, 50, 100)
    return F.max_pool2d(e.float().unsqueeze(1), 5, 1, 2)[:, 0] > 0
