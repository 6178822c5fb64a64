"""Trajectory error metrics for the `ATE within 1 %` acceptance check (the reference quotes ATE in
README.md:30 but ships no evaluator, so the alignment mode is this build's choice and is stated here).

ATE  = RMSE of the position error after a least-squares rigid alignment (Umeyama 1991, rotation + translation, optional
       scale) of the estimate onto the ground truth -- the KITTI/TUM convention.
RPE  = RMSE of the translational / rotational part of  (G_i^-1 G_{i+d})^-1 (X_i^-1 X_{i+d}).
Host-side numpy in float64: evaluation is off the timed path.
"""
import numpy as np


def umeyama(src, dst, with_scale=False):
    """Least-squares similarity (s, R, t) with  dst ~ s * R @ src + t  for (n,3) point sets."""
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / len(src)
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    var_s = (xs ** 2).sum() / len(src)
    s = float(np.trace(np.diag(D) @ S) / var_s) if with_scale and var_s > 0 else 1.0
    t = mu_d - s * R @ mu_s
    return s, R, t


def ate(est_pos, gt_pos, with_scale=False, align=True):
    """Absolute trajectory error (RMSE, metres) of (n,3) positions; returns (rmse, aligned_estimate)."""
    est_pos, gt_pos = np.asarray(est_pos, dtype=np.float64)[:, :3], np.asarray(gt_pos, dtype=np.float64)[:, :3]
    if align:
        s, R, t = umeyama(est_pos, gt_pos, with_scale)
        est_pos = s * est_pos @ R.T + t
    err = np.linalg.norm(est_pos - gt_pos, axis=1)
    return float(np.sqrt((err ** 2).mean())), est_pos


def _quat_to_mat(q):
    x, y, z, w = np.moveaxis(np.asarray(q, dtype=np.float64), -1, 0)
    n = x * x + y * y + z * z + w * w
    s = 2.0 / n
    return np.stack([np.stack([1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)], -1),
                     np.stack([s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)], -1),
                     np.stack([s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)], -1)], -2)


def rpe(est, gt, delta=1):
    """Relative pose error over `delta` frames for (n,7) [t, q xyzw] trajectories; returns (trans_rmse [m], rot_rmse [rad])."""
    est, gt = np.asarray(est, dtype=np.float64), np.asarray(gt, dtype=np.float64)

    def rel(X):
        R, t = _quat_to_mat(X[:, 3:]), X[:, :3]
        Ri, ti, Rj, tj = R[:-delta], t[:-delta], R[delta:], t[delta:]
        Rij = np.einsum('nji,njk->nik', Ri, Rj)
        tij = np.einsum('nji,nj->ni', Ri, tj - ti)
        return Rij, tij
    Re, te = rel(est)
    Rg, tg = rel(gt)
    dR = np.einsum('nji,njk->nik', Rg, Re)
    dt = np.einsum('nji,nj->ni', Rg, te - tg)
    ang = np.arccos(np.clip((np.trace(dR, axis1=1, axis2=2) - 1.0) / 2.0, -1.0, 1.0))
    return float(np.sqrt((dt ** 2).sum(1).mean())), float(np.sqrt((ang ** 2).mean()))
