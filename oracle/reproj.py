"""Oracle (test infrastructure): the sparse reprojection factor, PVGO's optional 5th residual.

CPU restatement (numpy) of
  * reference dense_ba.py:9-62     pixel2point
  * reference dense_ba.py:276-305  SparseReprojectionLoss.__init__ / __call__
  * reference pvgo.py:53-61        the hook in PoseVelGraph.forward (incl. the in-place `motion[0] = 0.1`, Appendix B)
  * PyPose (NOT in /root/reference, unpinned -- "parity unpinned"): pypose.function.geometry.reprojerr / point2pixel /
    homo2cart: pixels = homo2cart((extrinsics @ points) @ K^T), homo2cart divides by the last coordinate with its
    magnitude clamped to finfo.tiny.
The Jacobian is the exact left-perturbation derivative PyPose's autograd yields for these ops (pinned here against
finite differences, tests/test_oracle_cpu.py).
"""
import numpy as np

from . import lie


def pixel2point(pixels, depth, K4):
    """dense_ba.py:9-62.  pixels (...,N,2), depth (...,N), K4 = (fx, fy, cx, cy) -> (...,N,3)."""
    fx, fy, cx, cy = K4
    z = depth
    return np.stack([(pixels[..., 0] - cx) * z / fx, (pixels[..., 1] - cy) * z / fy, z], -1)


def homo2cart(c):
    tiny = np.finfo(c.dtype).tiny
    den = np.maximum(np.abs(c[..., -1:]), tiny)
    den = np.where(c[..., -1:] >= 0, den, -den)
    return c[..., :-1] / den


class SparseReprojection:
    """dense_ba.py:276-305 on numpy arrays.  bs = number of links, N = keypoints per link."""

    def __init__(self, points2d, depth, flow, fx, fy, cx, cy, rgb2imu_pose, dtype=np.float64):
        points2d = np.asarray(points2d)
        bs, N = points2d.shape[:2]
        col, row = points2d[..., 0].astype(np.int64), points2d[..., 1].astype(np.int64)      # idx = [b, y, x]
        b = np.arange(bs)[:, None]
        self.K4 = tuple(float(np.float32(v)) for v in (fx, fy, cx, cy))                        # K is float32 in the reference
        p2 = points2d.astype(dtype)
        self.point3d = pixel2point(p2, np.asarray(depth, dtype=dtype)[b, row, col], self.K4)
        self.target = np.asarray(flow, dtype=dtype).transpose(0, 2, 3, 1)[b, row, col, :] + p2
        self.N = N
        self.rgb2imu_pose = np.asarray(rgb2imu_pose, dtype=dtype)

    def camera_motion(self, motion):
        C = self.rgb2imu_pose[None]
        return lie.se3_mul(lie.se3_mul(lie.se3_inv(C), motion), C)

    def __call__(self, motion):
        """err (bs, N, 2) = point2pixel(point3d, K, T^-1) - target with T = rgb2imu^-1 motion rgb2imu."""
        fx, fy, cx, cy = self.K4
        Tinv = lie.se3_inv(self.camera_motion(motion))
        p = lie.se3_act(Tinv[:, None, :], self.point3d)
        hom = np.stack([fx * p[..., 0] + cx * p[..., 2], fy * p[..., 1] + cy * p[..., 2], p[..., 2]], -1)
        return homo2cart(hom) - self.target

    def jac_eta(self, motion):
        """d err / d eta (bs, N, 2, 6) for the left perturbation T <- Exp(eta) T of the camera-frame motion."""
        fx, fy, cx, cy = self.K4
        T = self.camera_motion(motion)
        Tinv = lie.se3_inv(T)
        p = lie.se3_act(Tinv[:, None, :], self.point3d)
        Rt = lie.quat_matrix(Tinv[:, 3:])[:, None]                                   # R_T^T
        P = self.point3d
        G = Rt @ np.concatenate([-np.broadcast_to(np.eye(3), P.shape[:-1] + (3, 3)), lie.skew(P)], -1)   # d p' / d eta
        x, y, z = p[..., 0], p[..., 1], p[..., 2]
        zero = np.zeros_like(z)
        Pi = np.stack([np.stack([fx / z, zero, -fx * x / (z * z)], -1), np.stack([zero, fy / z, -fy * y / (z * z)], -1)], -2)
        return Pi @ G


def link_motions(nodes, compat_first_motion=True):
    """pvgo.py:54-57: motion = X_k^-1 X_{k+1}; the reference then overwrites row 0 with the constant 0.1."""
    m = lie.se3_mul(lie.se3_inv(nodes[:-1]), nodes[1:])
    if compat_first_motion:
        m = m.copy()
        m[0] = 0.1
    return m


def residual(reproj, nodes, compat_first_motion=True):
    """(M, 2N) as pvgo.py:58-60."""
    return reproj(link_motions(nodes, compat_first_motion)).reshape(len(nodes) - 1, -1)


def jac_link(reproj, nodes, compat_first_motion=True):
    """J_link (M, 2N, 6): d reprojerr_k / d delta_{k+1};  d / d delta_k = -J_link.  Row 0 is constant under compat."""
    m = link_motions(nodes, compat_first_motion)
    Je = reproj.jac_eta(m)                                                           # (M, N, 2, 6)
    Cinv = lie.se3_inv(reproj.rgb2imu_pose)[None]
    Mad = lie.se3_adj(lie.se3_mul(Cinv, lie.se3_inv(nodes[:-1])))                    # Ad(C^-1) Ad(X_k^-1)
    J = (Je @ Mad[:, None]).reshape(len(nodes) - 1, -1, 6)
    if compat_first_motion:
        J[0] = 0.0
    return J
