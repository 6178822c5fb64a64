"""Oracle (test infrastructure): stereo scale recovery.

numpy restatement of reference dense_ba.py:88-176 (scale_from_disp_flow) with is_inside_image*
(:65-72, inclusive upper bound -- SURVEY Q6); ``depth`` selects the depth-input branch (:125-131), otherwise the
disparity branch TartanVO.py:163 uses.  float32 arithmetic like the reference; the 1-DoF normal equations are accumulated in
float64 (the HIP kernel reduces in a different order, tests use a tolerance)."""
import numpy as np

from . import lie


def scale_from_disp_flow(disp, flow, motion7, fx, fy, cx, cy, baseline, edge_mask=None, disp_th=1.0, depth=None):
    f32 = np.float32
    if depth is None:
        disp = np.asarray(disp, f32).reshape(disp.shape[-2:])
    flow = np.asarray(flow, f32)
    H, W = flow.shape[-2:]
    fx, fy, cx, cy, baseline = f32(fx), f32(fy), f32(cx), f32(cy), f32(baseline)
    u, v = np.meshgrid(np.arange(W, dtype=f32), np.arange(H, dtype=f32), indexing='xy')
    fu, fv = flow[0] + u, flow[1] + v
    inside = lambda a, n: np.logical_and(a >= 0, a <= n)
    flow_norm = np.sqrt(flow[0] * flow[0] + flow[1] * flow[1])
    mask = np.logical_and(np.logical_and(inside(fu, W), inside(fv, H)), flow_norm > 0)
    if edge_mask is not None:
        mask = np.logical_and(mask, np.asarray(edge_mask, bool))
    if depth is None:
        disp_mask = np.logical_and(inside(-disp + u, W), disp >= f32(disp_th))
        mask = np.logical_and(disp_mask, mask)
        with np.errstate(divide='ignore', invalid='ignore'):
            z = np.where(disp_mask, fx * baseline / disp, f32(0)).astype(f32)
    else:                                   # dense_ba.py:125-131
        depth = np.asarray(depth, f32).reshape(depth.shape[-2:])
        disp_mask = np.logical_and(depth <= fx * baseline, depth > 0)      # the reference calls it depth_mask
        mask = np.logical_and(disp_mask, mask)
        z = np.where(disp_mask, depth, f32(0)).astype(f32)
    # back-projection P = z * K^-1 [u v 1]
    Px = z * ((u - cx) / fx)
    Py = z * ((v - cy) / fy)
    Pz = z
    Tinv = lie.se3_inv(np.asarray(motion7, f32))
    R = lie.quat_matrix(Tinv[3:]).astype(f32)
    t = Tinv[:3]
    tn = t / max(np.linalg.norm(t), 1e-12)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], f32)
    a = K @ tn
    RP = np.stack([R[0, 0] * Px + R[0, 1] * Py + R[0, 2] * Pz,
                   R[1, 0] * Px + R[1, 1] * Py + R[1, 2] * Pz,
                   R[2, 0] * Px + R[2, 1] * Py + R[2, 2] * Pz])
    b0 = fx * RP[0] + cx * RP[2]
    b1 = fy * RP[1] + cy * RP[2]
    b2 = RP[2]
    M1 = a[2] * fu - a[0]
    w1 = b0 - b2 * fu
    M2 = a[2] * fv - a[1]
    w2 = b1 - b2 * fv
    m = mask
    MM = np.sum(M1[m].astype(np.float64) ** 2) + np.sum(M2[m].astype(np.float64) ** 2)
    Mw = np.sum(M1[m].astype(np.float64) * w1[m]) + np.sum(M2[m].astype(np.float64) * w2[m])
    with np.errstate(divide='ignore', invalid='ignore'):
        s = np.float64(1.0) / MM * Mw
    return f32(s), z, mask, disp_mask, (MM, Mw)
