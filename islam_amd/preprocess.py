"""Sample-dict preprocessing on the device (dataset half): what the reference does per
frame pair on DataLoader CPU workers with OpenCV -- CropCenter((448,640), fix_ratio=True) with its bilinear up-scale
(Datasets/utils.py:49-101, ResizeData :104-156), DownscaleFlow (:233-256, nearest x1/4 of the intrinsics layer),
Normalize(mean, std, keep_old=True) (:190-228), ToTensor -- as batched tensor ops on raw uint8 frames already in HBM.
Produces exactly the keys TartanVO.forward reads (SURVEY section 8b sample-dict contract).

The images are resized with OpenCV 4.7's INTER_LINEAR arithmetic on uint8 (11-bit fixed-point coefficients, its border rules;
restated for the tests in oracle/preprocess.py -- OpenCV itself is absent from the build container), the float32 intrinsics
layer with half-pixel-centre float interpolation like cv2.resize on float data."""
import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)          # train.py:82-83
IMAGENET_STD = (0.229, 0.224, 0.225)


def intrinsics_layer(w, h, fx, fy, ox, oy, device=None):
    """Datasets/utils.py:376-381 -> (2, h, w)."""
    ww, hh = torch.meshgrid(torch.arange(w, dtype=torch.float32, device=device),
                            torch.arange(h, dtype=torch.float32, device=device), indexing='xy')
    return torch.stack(((ww - ox + 0.5) / fx, (hh - oy + 0.5) / fy))


def crop_center_geometry(hh, ww, th=448, tw=640):
    """The integer geometry of CropCenter(fix_ratio=True): resized size (h, w), scales and crop origin (x1, y1)."""
    scale_h = max(1, float(th) / hh)
    scale_w = max(1, float(tw) / ww)
    if scale_h > 1 or scale_w > 1:
        scale_h = max(scale_h, scale_w)
        scale_w = max(scale_h, scale_w)
        w, h = int(round(ww * scale_w)), int(round(hh * scale_h))
    else:
        w, h = ww, hh
    return h, w, int((w - tw) / 2), int((h - th) / 2)


def _cv_coeffs(dn, sn, clamp_weights):
    """cv2.resize INTER_LINEAR coordinates (imgproc/resize.cpp): source index and float32 fraction per destination index;
    columns zero the fraction at the borders, rows keep it and clamp the two row indices."""
    scale = 1.0 / (float(dn) / float(sn))
    f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_weights:
        lo, hi = s < 0, s >= sn - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, sn - 1, s))
    return s, f


def resize_linear_u8(x, h, w):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) on a (B,C,H,W) uint8 batch, on the device, in OpenCV's
    fixed-point arithmetic: D = S[sx]*a0 + S[sx+1]*a1, dst = (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2."""
    H, W = x.shape[-2:]
    if (H, W) == (h, w):
        return x
    dev = x.device
    sx, fx = _cv_coeffs(w, W, True)
    sy, fy = _cv_coeffs(h, H, False)
    q = lambda f: torch.from_numpy(np.rint(f * np.float32(2048)).astype(np.int32)).to(dev)
    a0, a1, b0, b1 = q(np.float32(1) - fx), q(fx), q(np.float32(1) - fy), q(fy)
    ix0 = torch.from_numpy(sx).to(dev)
    ix1 = torch.from_numpy(np.minimum(sx + 1, W - 1)).to(dev)
    iy0 = torch.from_numpy(np.clip(sy, 0, H - 1)).to(dev)
    iy1 = torch.from_numpy(np.clip(sy + 1, 0, H - 1)).to(dev)
    xi = x.to(torch.int32)
    hres = xi.index_select(-1, ix0) * a0 + xi.index_select(-1, ix1) * a1
    r0, r1 = hres.index_select(-2, iy0) >> 4, hres.index_select(-2, iy1) >> 4
    out = (((r0 * b0[:, None]) >> 16) + ((r1 * b1[:, None]) >> 16) + 2) >> 2
    return out.clamp_(0, 255).to(torch.uint8)


def _resize_crop(x, h, w, x1, y1, th, tw):
    if x.shape[-2:] != (h, w):
        x = F.interpolate(x, size=(h, w), mode='bilinear', align_corners=False)
    return x[..., y1:y1 + th, x1:x1 + tw]


def make_sample(img0, img1, img0_r, intrinsic_calib, extrinsic, datatype, size=(448, 640), links=None, dts=None):
    """img0, img1, img0_r: (B,H,W,3) or (B,3,H,W) uint8 BGR frames on the device (left t, left t+1, right t);
    intrinsic_calib (B,4) = fx, fy, cx, cy at the raw resolution; extrinsic (B,7) right->left.  Returns the sample dict."""
    th, tw = size
    dev = img0.device

    def chw(x):
        return x.permute(0, 3, 1, 2) if x.shape[-1] == 3 else x
    a, b, r = chw(img0), chw(img1), chw(img0_r)
    B, _, hh, ww = a.shape
    h, w, x1, y1 = crop_center_geometry(hh, ww, th, tw)
    calib = intrinsic_calib.to(dev, torch.float32).clone()
    sw, sh = float(w) / ww, float(h) / hh                         # ResizeData rescales the intrinsics (:151-155)
    calib[:, 0] *= sw
    calib[:, 2] *= sw
    calib[:, 1] *= sh
    calib[:, 3] *= sh
    # the intrinsics layer is built at the raw size, then resized and cropped like an image (TrajFolderDataset.py:497-500)
    raw = intrinsic_calib.to(dev, torch.float32)
    layer = torch.stack([intrinsics_layer(ww, hh, *raw[i].tolist(), device=dev) for i in range(B)])
    layer = _resize_crop(layer, h, w, x1, y1, th, tw)[..., ::4, ::4]      # DownscaleFlow: INTER_NEAREST at 1/4
    calib[:, 2] -= x1                                             # CropCenter (:98-100)
    calib[:, 3] -= y1
    mean = torch.tensor(IMAGENET_MEAN, device=dev).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=dev).view(1, 3, 1, 1)
    out = {}
    for name, x in (('img0', a), ('img1', b), ('img0_r', r)):
        if x.dtype == torch.uint8:                                # cv2.resize on the raw uint8 frame, then crop, then /255
            x = resize_linear_u8(x, h, w)[..., y1:y1 + th, x1:x1 + tw].float() / 255.0
        else:
            x = _resize_crop(x.float(), h, w, x1, y1, th, tw) / 255.0
        out[name] = x.contiguous()
        out[name + '_norm'] = ((x - mean) / std).contiguous()     # Normalize(keep_old=True)
    out.update(intrinsic=layer.contiguous(), intrinsic_calib=calib.cpu(), extrinsic=extrinsic.cpu().float(),
               datatype=list(datatype))
    if links is not None:
        out['link'] = links
    if dts is not None:
        out['dt'] = dts
    return out
