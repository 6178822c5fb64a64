"""The reference's ``dense_ba`` module surface (reference dense_ba.py) on the islam_amd back-end.

  * scale_from_disp_flow (dense_ba.py:88-176): single-sample wrapper of the batched HIP reduction (TartanVO.stereo_scale);
  * SparseReprojectionLoss (dense_ba.py:276-305): holds the back-projected keypoints / flow targets that run_pvgo hands to
    the HIP LM loop as the 5th residual (islam_pvgo_run_chain_reproj); ``__call__`` evaluates the residual with LieTensor
    ops on whatever device its tensors live on (used for inspection and by the tests);
  * DenseReprojectionLoss (dense_ba.py:179-273): the per-frame mean L1 reprojection error over the masked pixels.  The
    reference cannot feed it to run_pvgo (it has no ``.N``, pvgo.py:131), so it is evaluation-only here as well;
  * pixel2point / proj / is_inside_image helpers with the reference's semantics;
  * sample_keypoints: device-side choice of ``n`` masked pixels per frame (the reference only declares
    ``--reproj-points``, arguments.py:62; no sampler ships with it).
"""
import torch

from . import lietensor as pp


def pixel2point(pixels, depth, intrinsics):
    """dense_ba.py:9-62: pixels (...,N,2), depth (...,N), intrinsics (...,3,3) -> camera-frame points (...,N,3)."""
    assert pixels.size(-1) == 2, "Pixels shape incorrect"
    assert depth.size(-1) == pixels.size(-2), "Depth shape does not match pixels"
    assert intrinsics.size(-1) == intrinsics.size(-2) == 3, "Intrinsics shape incorrect."
    fx, fy = intrinsics[..., 0, 0], intrinsics[..., 1, 1]
    cx, cy = intrinsics[..., 0, 2], intrinsics[..., 1, 2]
    assert not torch.any(fx == 0), "fx Cannot contain zero"
    assert not torch.any(fy == 0), "fy Cannot contain zero"
    z = depth
    return torch.stack([((pixels[..., 0] - cx) * z) / fx, ((pixels[..., 1] - cy) * z) / fy, z], dim=-1)


def is_inside_image_1D(u, width):
    return torch.logical_and(u >= 0, u <= width)


def is_inside_image(uv, width, height):
    if len(uv.shape) == 3:
        return torch.logical_and(is_inside_image_1D(uv[0, ...], width), is_inside_image_1D(uv[1, ...], height))
    return torch.logical_and(is_inside_image_1D(uv[:, 0, ...], width), is_inside_image_1D(uv[:, 1, ...], height))


def proj(x, return_mask=False):
    """dense_ba.py:71-85: normalised image coordinates; the mask keeps z > 0.1 and |x/z|, |y/z| <= 1."""
    if not return_mask:
        return x / x[..., -1]
    mask = x[..., -1:] > 0.1
    p = torch.where(mask, x / x[..., -1:], torch.zeros_like(x))
    mask = torch.where(mask, (p[..., 0:1] >= -1) & (p[..., 0:1] <= 1) & (p[..., 1:2] >= -1) & (p[..., 1:2] <= 1),
                       torch.zeros_like(mask))
    p = torch.where(mask, p, torch.zeros_like(p))
    return p, mask.squeeze()


def scale_from_disp_flow(disp, flow, motion, fx, fy, cx, cy, baseline, depth=None, mask=None, disp_th=1):
    """dense_ba.py:88-176 for ONE sample: disp (1,H,W) or (H,W), flow (2,H,W), motion SE3 (7) or se3 (6).
    Returns (s (1,), z (H,W), mask (H,W), depth_mask (H,W)); differentiable w.r.t. ``motion``."""
    from .TartanVO import stereo_scale
    dev = flow.device
    if isinstance(motion, pp.LieTensor):
        T = motion if motion.shape[-1] == 7 else motion.Exp()
    else:
        T = pp.SE3(motion) if motion.shape[-1] == 7 else pp.se3(motion).Exp()
    T = pp.SE3(T.tensor().reshape(1, 7))
    H, W = flow.shape[-2:]
    f32 = lambda v: torch.as_tensor(v, dtype=torch.float32).reshape(-1)
    intr = torch.cat([f32(fx), f32(fy), f32(cx), f32(cy)]).reshape(1, 4)
    edge = None if mask is None else mask.reshape(1, H, W)
    if depth is not None:        # dense_ba.py:125-131: a depth map replaces the disparity, valid where 0 < depth <= fx*baseline
        s, z, m, dm = stereo_scale(depth.reshape(1, 1, H, W), flow.reshape(1, 2, H, W), T, intr, f32(baseline), edge, None,
                                   depth_input=True)
    else:
        s, z, m, dm = stereo_scale(disp.reshape(1, 1, H, W), flow.reshape(1, 2, H, W), T, intr, f32(baseline), edge, f32(disp_th))
    if int(m.sum()) < 500:
        print('Warning! mask contains too less points!', int(m.sum()))          # dense_ba.py:133-134
    return s.reshape(1), z[0], m[0], dm[0]


def sample_keypoints(mask, n, generator=None):
    """Choose ``n`` pixels per frame among the True entries of mask (B,H,W), uniformly without replacement (with
    replacement when a frame has fewer than n).  Returns float (B,n,2) [u, v] pixel coordinates on mask's device."""
    B, H, W = mask.shape
    flat = mask.reshape(B, -1).to(torch.float32)
    few = flat.sum(1) < n
    if bool(few.any()):
        flat = torch.where(flat.sum(1, keepdim=True) > 0, flat, torch.ones_like(flat))
    idx = torch.multinomial(flat, n, replacement=bool(few.any()), generator=generator)
    return torch.stack([(idx % W), (idx // W)], -1).to(torch.float32)


class SparseReprojectionLoss:
    def __init__(self, points2d, depth, flow, fx, fy, cx, cy, rgb2imu_pose, device='cuda:0'):
        assert len(flow.shape) == 4            # (batch, channel, height, width)
        assert len(depth.shape) == 3           # (batch, height, width)
        assert len(points2d.shape) == 3        # (batch, N, 2)
        bs, N = points2d.shape[:2]
        points2d = points2d.to(device)
        b = torch.arange(bs, device=device).view(bs, 1).expand(bs, N)
        row, col = points2d[..., 1].to(torch.int64), points2d[..., 0].to(torch.int64)        # idx = [b, y, x] (:287)
        self.K = torch.tensor([fx, 0, cx, 0, fy, cy, 0, 0, 1], dtype=torch.float32).view(3, 3).to(device)
        depth, flow = depth.to(device), flow.to(device)
        self.point3d = pixel2point(points2d, depth[b, row, col].view(bs, N), self.K)
        self.target = flow.permute(0, 2, 3, 1)[b, row, col, :].view(bs, N, 2) + points2d
        self.N = N
        self.rgb2imu_pose = rgb2imu_pose.to(device)
        self.compat_first_motion = True        # run_pvgo replicates pvgo.py:57 `motion[0] = 0.1` unless cleared

    def __call__(self, motion):
        """(bs, N, 2) = reprojerr(point3d, target, K, T^-1, reduction='none') with T = rgb2imu^-1 motion rgb2imu."""
        C = self.rgb2imu_pose
        T = C.Inv() @ motion @ C
        Tinv = T.Inv()
        dt = pp._plain(Tinv).dtype
        P = self.point3d.to(dt)
        p = pp.SE3(pp._plain(Tinv).unsqueeze(-2)).Act(P)
        K = self.K.to(dt)
        hom = p @ K.mT
        tiny = torch.finfo(dt).tiny
        den = hom[..., -1:].abs().clamp(min=tiny)
        den = torch.where(hom[..., -1:] >= 0, den, -den)
        return hom[..., :-1] / den - self.target.to(dt)


class DenseReprojectionLoss:
    def __init__(self, depth, flow, fx, fy, cx, cy, mask, rgb2imu_pose, device='cuda:0'):
        assert len(flow.shape) == 4
        assert len(depth.shape) == 3
        assert mask is None or len(mask.shape) == 3
        H, W = flow.shape[-2:]
        self.z, self.flow = depth.to(device), flow.to(device)
        self.mask = torch.ones_like(self.z, dtype=torch.bool) if mask is None else mask.to(device)
        self.rgb2imu_pose = rgb2imu_pose.to(device)
        u, v = torch.meshgrid(torch.linspace(0, W - 1, W, device=device), torch.linspace(0, H - 1, H, device=device), indexing='xy')
        self.uv = torch.stack([u, v])
        self.K4 = (float(fx), float(fy), float(cx), float(cy))

    def __call__(self, motion):
        """dense_ba.py:209-236: mean over the masked pixels of |reproj - (flow + uv)|_1 per frame -> (B,)."""
        fx, fy, cx, cy = self.K4
        C = self.rgb2imu_pose
        Tinv = (C.Inv() @ motion @ C).Inv()
        dt = pp._plain(Tinv).dtype
        u, v = self.uv[0].to(dt), self.uv[1].to(dt)
        z = self.z.to(dt)
        P = torch.stack([z * (u - cx) / fx, z * (v - cy) / fy, z], -1)                       # (B,H,W,3)
        B, H, W = z.shape
        Pt = pp.SE3(pp._plain(Tinv).view(B, 1, 7)).Act(P.view(B, H * W, 3)).view(B, H, W, 3)
        p, rmask = proj(Pt, return_mask=True)
        mask = self.mask & rmask.view(B, H, W)
        ru = fx * p[..., 0] + cx * p[..., 2] - (self.flow[:, 0].to(dt) + u)
        rv = fy * p[..., 1] + cy * p[..., 2] - (self.flow[:, 1].to(dt) + v)
        l1 = ru.abs() + rv.abs()
        return torch.stack([l1[i][mask[i]].mean() for i in range(B)])
