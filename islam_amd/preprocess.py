"""Sample-dict preprocessing on the device (dataset half): what the reference does per
frame pair on DataLoader CPU workers with OpenCV -- CropCenter((448,640), fix_ratio=True) with its bilinear up-scale
(Datasets/utils.py:49-101, ResizeData :104-156), DownscaleFlow (:233-256, nearest x1/4 of the intrinsics layer),
Normalize(mean, std, keep_old=True) (:190-228), ToTensor -- as batched tensor ops on raw uint8 frames already in HBM.
Produces exactly the keys TartanVO.forward reads (SURVEY section 8b sample-dict contract).

OpenCV is absent from the build container: INTER_LINEAR is restated as half-pixel-centre bilinear interpolation
(torch ``align_corners=False``), evaluated in float32 (OpenCV uses 11-bit fixed point on uint8: <= 1/255 apart)."""
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
        x = x.permute(0, 3, 1, 2) if x.shape[-1] == 3 else x
        return x.float()
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
        x = _resize_crop(x, h, w, x1, y1, th, tw) / 255.0
        out[name] = x.contiguous()
        out[name + '_norm'] = ((x - mean) / std).contiguous()     # Normalize(keep_old=True)
    out.update(intrinsic=layer.contiguous(), intrinsic_calib=calib.cpu(), extrinsic=extrinsic.cpu().float(),
               datatype=list(datatype))
    if links is not None:
        out['link'] = links
    if dts is not None:
        out['dt'] = dts
    return out
