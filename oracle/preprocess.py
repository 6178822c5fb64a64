"""Oracle (test infrastructure): the DataLoader-side preprocessing of a frame pair (reference Datasets/utils.py:49-156 CropCenter +
ResizeData, :190-228 Normalize, :233-256 DownscaleFlow, :376-381 make_intrinsics_layer; TrajFolderDataset.py:497-500), with
cv2.resize restated from OpenCV 4.7 (imgproc/resize.cpp; opencv-python 4.7.0.68 is what environment.yml:149 pins and is NOT
installable here: **parity unpinned**).

cv2.resize(img, (w, h), interpolation=INTER_LINEAR):
  coordinates   fx = (float)((dx + 0.5) * scale_x - 0.5), scale_x = 1 / ((double)dw / sw);  sx = floor(fx), fx -= sx;
                sx < 0 -> sx = 0, fx = 0;  sx >= sw - 1 -> sx = sw - 1, fx = 0   (same for rows, except that rows keep fy and only
                CLAMP the two row indices to [0, sh - 1])
  uint8 images  11-bit fixed point: alpha = cvRound((1 - fx, fx) * 2048) as short (round half to even), horizontal pass
                D = S[sx] * a0 + S[sx + 1] * a1 (int), vertical pass dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
  float32       plain float interpolation: D = S[sx] * (1 - fx) + S[sx + 1] * fx, dst = S0 * (1 - fy) + S1 * fy
cv2.resize(..., fx=1/4, fy=1/4, interpolation=INTER_NEAREST): source index floor(dx * 4) -> [::4, ::4]."""
import numpy as np


def _coeffs(dn, sn, clamp_weights):
    """Per destination index: source index s0 (the second tap is min(s0 + 1, sn - 1)) and the float32 fraction."""
    scale = 1.0 / (float(dn) / float(sn))
    f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_weights:                                   # columns: the fraction is zeroed at the borders
        lo, hi = s < 0, s >= sn - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, sn - 1, s))
    return s, f


def resize_linear_u8(img, h, w):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for (H, W, C) uint8."""
    H, W = img.shape[:2]
    if (H, W) == (h, w):
        return img.copy()
    sx, fx = _coeffs(w, W, True)
    sy, fy = _coeffs(h, H, False)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
    x = img.astype(np.int64)
    x1 = np.minimum(sx + 1, W - 1)
    hres = x[:, sx] * a0[None, :, None] + x[:, x1] * a1[None, :, None]                 # (H, w, C)
    r0, r1 = np.clip(sy, 0, H - 1), np.clip(sy + 1, 0, H - 1)
    out = (((b0[:, None, None] * (hres[r0] >> 4)) >> 16) + ((b1[:, None, None] * (hres[r1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_linear_f32(img, h, w):
    """cv2.resize for (H, W, C) float32."""
    H, W = img.shape[:2]
    if (H, W) == (h, w):
        return img.copy()
    f32 = np.float32
    sx, fx = _coeffs(w, W, True)
    sy, fy = _coeffs(h, H, False)
    x1 = np.minimum(sx + 1, W - 1)
    hres = (img[:, sx] * (f32(1) - fx)[None, :, None] + img[:, x1] * fx[None, :, None]).astype(f32)
    r0, r1 = np.clip(sy, 0, H - 1), np.clip(sy + 1, 0, H - 1)
    return (hres[r0] * (f32(1) - fy)[:, None, None] + hres[r1] * fy[:, None, None]).astype(f32)


def crop_center_geometry(hh, ww, th=448, tw=640):
    """Datasets/utils.py:66-89 (fix_ratio=True, scale_w=1): resized size (h, w) and crop origin (x1, y1)."""
    scale_h, scale_w = max(1, float(th) / hh), max(1, float(tw) / ww)
    if scale_h > 1 or scale_w > 1:
        scale_h = max(scale_h, scale_w)
        scale_w = max(scale_h, scale_w)
        w, h = int(round(ww * scale_w)), int(round(hh * scale_h))
    else:
        w, h = ww, hh
    return h, w, int((w - tw) / 2), int((h - th) / 2)


def make_sample(img0, img1, img0_r, calib, size=(448, 640), mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """One frame pair: (H, W, 3) uint8 BGR images, calib = (fx, fy, cx, cy) at the raw size -> the tensors TartanVO reads."""
    th, tw = size
    hh, ww = img0.shape[:2]
    h, w, x1, y1 = crop_center_geometry(hh, ww, th, tw)
    f32 = np.float32
    fx, fy, ox, oy = [f32(v) for v in calib]
    u, v = np.meshgrid(np.arange(ww, dtype=f32), np.arange(hh, dtype=f32))
    layer = np.stack(((u - ox + f32(0.5)) / fx, (v - oy + f32(0.5)) / fy), -1).astype(f32)          # utils.py:376-381
    layer = resize_linear_f32(layer, h, w)[y1:y1 + th, x1:x1 + tw][::4, ::4]
    out = {'intrinsic': layer.transpose(2, 0, 1).copy()}
    c = np.array(calib, dtype=np.float64)
    sw, sh = float(w) / ww, float(h) / hh
    c = c * np.array([sw, sh, sw, sh]) - np.array([0, 0, x1, y1])
    out['intrinsic_calib'] = c.astype(f32)
    m, s = np.array(mean, f32).reshape(3, 1, 1), np.array(std, f32).reshape(3, 1, 1)
    for name, im in (('img0', img0), ('img1', img1), ('img0_r', img0_r)):
        x = resize_linear_u8(im, h, w)[y1:y1 + th, x1:x1 + tw].transpose(2, 0, 1).astype(f32) / f32(255)
        out[name] = x
        out[name + '_norm'] = ((x - m) / s).astype(f32)
    return out
