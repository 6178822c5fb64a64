"""Shared builders of seeded test problems (tests only; may import the oracle)."""
import numpy as np

from islam_amd import synthetic
from oracle import imu as oimu


def chain_problem(n_frames, seed=11, **kw):
    """Synthetic car trajectory -> run_pvgo inputs, IMU quantities from the oracle integrator."""
    tr = synthetic.car_trajectory(n_frames, seed=seed, **kw)
    F = n_frames
    pos, rot, vel = oimu.integrate(tr['accels'], tr['gyros'], tr['imu_dts'], tr['rgb2imu_sync'], 0, F - 1, tr['init'],
                                   tr['gravity'], False)
    dpos, drot, dvel = oimu.integrate(tr['accels'], tr['gyros'], tr['imu_dts'], tr['rgb2imu_sync'], 0, F - 1, tr['init'],
                                      tr['gravity'], True)
    return synthetic.pvgo_problem_from_deltas(tr, drot, dpos, dvel, pos, rot, vel), tr


def se3_log_err(X, Xref):
    from oracle import lie
    d = lie.se3_log(lie.se3_mul(lie.se3_inv(Xref), X))
    return np.linalg.norm(d, axis=-1)


def reproj_inputs(tr, K, T_IL, H=112, W=160, K4=(80.0, 80.0, 80.0, 56.0), seed=0, noise_px=0.3):
    """Keypoints, depth and flow maps for a SparseReprojectionLoss over every link of trajectory `tr`: the flow at the
    keypoints is the true reprojection under the ground-truth motion (camera frame) plus pixel noise."""
    from oracle import lie, reproj as orp
    rng = np.random.default_rng(seed)
    gt = np.concatenate([tr['gt_pos'], tr['gt_quat']], 1)
    M = len(gt) - 1
    T_IL = np.asarray(T_IL, dtype=np.float64)
    mot = lie.se3_mul(lie.se3_inv(gt[:-1]), gt[1:])
    cam = lie.se3_mul(lie.se3_mul(lie.se3_inv(T_IL)[None], mot), T_IL[None])
    pts2d = np.stack([rng.integers(0, W, (M, K)), rng.integers(0, H, (M, K))], -1).astype(np.float64)
    depth = rng.uniform(5, 40, (M, H, W))
    flow = np.zeros((M, 2, H, W))
    b = np.arange(M)[:, None]
    col, row = pts2d[..., 0].astype(int), pts2d[..., 1].astype(int)
    fx, fy, cx, cy = K4
    P = orp.pixel2point(pts2d, depth[b, row, col], K4)
    p = lie.se3_act(lie.se3_inv(cam)[:, None, :], P)
    uv = np.stack([fx * p[..., 0] / p[..., 2] + cx, fy * p[..., 1] / p[..., 2] + cy], -1)
    flow[b, :, row, col] = uv - pts2d + rng.normal(0, noise_px, uv.shape)
    return dict(points2d=pts2d, depth=depth, flow=flow, fx=fx, fy=fy, cx=cx, cy=cy, rgb2imu_pose=T_IL)
