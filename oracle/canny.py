"""Oracle (test infrastructure): the edge mask of TartanVO.forward (reference TartanVO.py:145-155).

    img0_np = (img0.cpu().numpy().transpose(0, 2, 3, 1) * 255).astype(np.uint8)
    im = cv2.resize(img0_np[i], None, fx=1/4, fy=1/4); e = cv2.Canny(im, 50, 100)
    e = cv2.dilate(e, np.ones((5, 5), np.uint8)); e = e > 0

The arithmetic lives in opencv-python 4.7.0.68 (reference environment.yml:149), which is NOT under /root/reference
and not installable here: **parity unpinned**.  This file restates OpenCV 4.7's published algorithm, one numpy
function per OpenCV call, in OpenCV's own order of operations (sequential stack-based hysteresis -- deliberately a
different formulation from the HIP kernel's in-LDS fixed-point iteration):

  resize (imgproc/resize.cpp, INTER_LINEAR on CV_8U, scale exactly 1/4): source coordinate of destination x is
      (x + 0.5) * 4 - 0.5 = 4x + 1.5, i.e. taps 4x+1, 4x+2 with weight 1/2 each -> fixed point 1024 / 2048
      (INTER_RESIZE_COEF_BITS = 11); HResize accumulates 1024 * (a + b) as int, VResizeLinear<uchar> computes
      ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2  =  (a + b + c + d + 2) >> 2.
  Canny (imgproc/canny.cpp, aperture 3, L2gradient = false, 8-bit 3-channel input):
      low = cvFloor(50), high = cvFloor(100); Sobel 3x3 dx, dy as CV_16S with BORDER_REPLICATE per channel;
      |dx| + |dy| per channel, the FIRST channel holding the maximum wins (strict > when scanning k = 1..cn-1);
      the magnitude buffer is zero outside the image;
      non-maximum suppression only where mag > low, sector by fixed-point tangent TG22 = 13573 (= tan 22.5 * 2^15):
          y = |dy| << 15, tg22x = |dx| * TG22, tg67x = tg22x + (|dx| << 16)
          y < tg22x : keep if m > left and m >= right
          y > tg67x : keep if m > up   and m >= down
          else      : s = sign(dx ^ dy); keep if m > up[x - s] and m > down[x + s]
      kept and m > high -> edge (2, pushed on the stack); kept otherwise -> candidate (0); everything else 1;
      hysteresis: pop, promote the 8 neighbours that are 0 to 2 and push them; output 255 where map == 2.
  dilate (imgproc/morph.cpp, 5x5 ones, anchor centre, BORDER_CONSTANT with the morphology default = never wins).
"""
import numpy as np

TG22 = 13573


def to_u8(img0):
    """(B,3,H,W) float32 in [0,1] -> (B,H,W,3) uint8, `(img * 255).astype(np.uint8)` (float32 product, truncation)."""
    x = np.asarray(img0, dtype=np.float32).transpose(0, 2, 3, 1) * np.float32(255)
    return x.astype(np.uint8)


def resize_quarter(im):
    """cv2.resize(im, None, fx=1/4, fy=1/4) for (H,W,C) uint8 with H, W multiples of 4."""
    H, W = im.shape[:2]
    assert H % 4 == 0 and W % 4 == 0
    x = im.astype(np.int32)
    hres = 1024 * (x[:, 1::4] + x[:, 2::4])                     # HResizeLinear: int accumulators, alpha = (1024, 1024)
    s0, s1 = hres[1::4], hres[2::4]
    out = (((1024 * (s0 >> 4)) >> 16) + ((1024 * (s1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def sobel_3x3(ch):
    """cv::Sobel(src, CV_16S, ., ., 3, 1, 0, BORDER_REPLICATE) for one (H,W) channel: dx, dy as int32."""
    p = np.pad(ch.astype(np.int32), 1, mode='edge')
    dx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    dy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    return dx, dy


def canny(im, low_thresh=50, high_thresh=100):
    """cv2.Canny(im, 50, 100) for (H,W,C) uint8 -> (H,W) uint8 in {0, 255}."""
    if im.ndim == 2:
        im = im[:, :, None]
    H, W, C = im.shape
    low, high = int(np.floor(min(low_thresh, high_thresh))), int(np.floor(max(low_thresh, high_thresh)))
    dxs, dys = zip(*(sobel_3x3(im[:, :, c]) for c in range(C)))
    dx, dy = dxs[0].copy(), dys[0].copy()
    mag = np.abs(dx) + np.abs(dy)
    for c in range(1, C):
        n = np.abs(dxs[c]) + np.abs(dys[c])
        take = n > mag
        mag, dx, dy = np.where(take, n, mag), np.where(take, dxs[c], dx), np.where(take, dys[c], dy)
    mp = np.pad(mag, 1)                                           # zero border rows / columns of the magnitude buffer
    pmap = np.ones((H + 2, W + 2), dtype=np.uint8)                # 1 = cannot be an edge (incl. the border ring)
    stack = []
    for i in range(H):
        for j in range(W):
            m = int(mag[i, j])
            if m <= low:
                continue
            xs, ys = int(dx[i, j]), int(dy[i, j])
            x, y = abs(xs), abs(ys) << 15
            tg22x = x * TG22
            a, b = i + 1, j + 1
            if y < tg22x:
                keep = m > mp[a, b - 1] and m >= mp[a, b + 1]
            else:
                tg67x = tg22x + (x << 16)
                if y > tg67x:
                    keep = m > mp[a - 1, b] and m >= mp[a + 1, b]
                else:
                    s = -1 if (xs ^ ys) < 0 else 1
                    keep = m > mp[a - 1, b - s] and m > mp[a + 1, b + s]
            if keep:
                if m > high:
                    pmap[a, b] = 2
                    stack.append((a, b))
                else:
                    pmap[a, b] = 0
    while stack:
        a, b = stack.pop()
        for da in (-1, 0, 1):
            for db in (-1, 0, 1):
                if (da or db) and pmap[a + da, b + db] == 0:
                    pmap[a + da, b + db] = 2
                    stack.append((a + da, b + db))
    return np.where(pmap[1:-1, 1:-1] == 2, 255, 0).astype(np.uint8)


def dilate(e, k=5):
    """cv2.dilate(e, np.ones((k, k), np.uint8))."""
    r = k // 2
    p = np.pad(e, r)
    out = np.zeros_like(e)
    for a in range(k):
        for b in range(k):
            out = np.maximum(out, p[a:a + e.shape[0], b:b + e.shape[1]])
    return out


def edge_mask(img0, downscale=True):
    """TartanVO.py:145-155: (B,3,H,W) float32 -> (B,H/4,W/4) bool."""
    out = []
    for im in to_u8(img0):
        if downscale:
            im = resize_quarter(im)
        out.append(dilate(canny(im, 50, 100), 5) > 0)
    return np.stack(out)
