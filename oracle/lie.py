"""Oracle (test infrastructure): SO(3)/SE(3) closed forms in numpy.

Restates the formulas PyPose uses for the LieTensor operations the reference
calls (pvgo.py:36-51,70-76,116-118; Datasets/transformation.py:72-124;
imu_integrator.py:151; dense_ba.py:142-143).  PyPose itself is absent from
/root/reference, so this follows its published algorithm
(pypose/lietensor/operation.py, v0.6.x): quaternion order [x,y,z,w], SE3 =
[t, q], tangent order [rho, phi], eps-thresholded Taylor branches.

All functions broadcast over leading dimensions and keep the input dtype.
"""
import numpy as np


def _eps(x):
    return np.finfo(x.dtype).eps


def skew(v):
    v = np.asarray(v)
    z = np.zeros_like(v[..., 0])
    return np.stack([
        np.stack([z, -v[..., 2], v[..., 1]], -1),
        np.stack([v[..., 2], z, -v[..., 0]], -1),
        np.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)


# --------------------------------------------------------------------- SO3
def so3_exp(phi):
    """so3 -> SO3 quaternion.  PyPose so3_Exp."""
    phi = np.asarray(phi)
    th = np.linalg.norm(phi, axis=-1, keepdims=True)
    th2 = th * th
    th4 = th2 * th2
    big = th > _eps(phi)
    ths = np.where(big, th, 1.0)
    imag = np.where(big, np.sin(0.5 * ths) / ths, 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4)
    real = np.where(big, np.cos(0.5 * ths), 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4)
    return np.concatenate([phi * imag, real], -1).astype(phi.dtype)


def so3_log(q):
    """SO3 quaternion -> so3.  PyPose SO3_Log (atan, not atan2)."""
    q = np.asarray(q)
    v, w = q[..., :3], q[..., 3:]
    vn = np.linalg.norm(v, axis=-1, keepdims=True)
    big = vn > _eps(q)
    vns = np.where(big, vn, 1.0)
    with np.errstate(divide='ignore', invalid='ignore'):
        f = np.where(big, 2.0 * np.arctan(vns / w) / vns, 2.0 / w - (2.0 / 3.0) * (vn * vn) / (w * w * w))
    return (f * v).astype(q.dtype)


def quat_mul(a, b):
    """Hamilton product, [x,y,z,w] storage (SO3_Mul)."""
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz], -1)


def quat_inv(q):
    return np.concatenate([-q[..., :3], q[..., 3:]], -1)


def quat_act(q, p):
    """Rotate points (SO3_Act): p + w*(2u x p) + u x (2u x p)."""
    u, w = q[..., :3], q[..., 3:]
    uv = 2.0 * cross(u, p)
    return p + w * uv + cross(u, uv)


def quat_matrix(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
        np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
        np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def so3_Jl(phi):
    phi = np.asarray(phi)
    K = skew(phi)
    th = np.linalg.norm(phi, axis=-1)[..., None, None]
    big = th > _eps(phi)
    ths = np.where(big, th, 1.0)
    c1 = np.where(big, (1 - np.cos(ths)) / ths ** 2, 0.5)
    c2 = np.where(big, (ths - np.sin(ths)) / ths ** 3, 1.0 / 6.0)
    I = np.eye(3, dtype=phi.dtype)
    return I + c1 * K + c2 * (K @ K)


def so3_Jl_inv(phi):
    phi = np.asarray(phi)
    K = skew(phi)
    th = np.linalg.norm(phi, axis=-1)[..., None, None]
    big = th > _eps(phi)
    ths = np.where(big, th, 1.0)
    c2 = np.where(big, (1 - ths * np.cos(0.5 * ths) / (2 * np.sin(0.5 * ths))) / ths ** 2, 1.0 / 12.0)
    I = np.eye(3, dtype=phi.dtype)
    return I - 0.5 * K + c2 * (K @ K)


# --------------------------------------------------------------------- SE3
def se3_exp(xi):
    xi = np.asarray(xi)
    rho, phi = xi[..., :3], xi[..., 3:]
    q = so3_exp(phi)
    t = (so3_Jl(phi) @ rho[..., None])[..., 0]
    return np.concatenate([t, q], -1).astype(xi.dtype)


def se3_log(X):
    X = np.asarray(X)
    t, q = X[..., :3], X[..., 3:]
    phi = so3_log(q)
    rho = (so3_Jl_inv(phi) @ t[..., None])[..., 0]
    return np.concatenate([rho, phi], -1).astype(X.dtype)


def se3_mul(X, Y):
    tx, qx = X[..., :3], X[..., 3:]
    ty, qy = Y[..., :3], Y[..., 3:]
    return np.concatenate([tx + quat_act(qx, ty), quat_mul(qx, qy)], -1)


def se3_inv(X):
    t, q = X[..., :3], X[..., 3:]
    qi = quat_inv(q)
    return np.concatenate([-quat_act(qi, t), qi], -1)


def se3_act(X, p):
    return quat_act(X[..., 3:], p) + X[..., :3]


def se3_adj(X):
    """Ad(X) = [[R, [t]x R],[0, R]] in [rho, phi] order."""
    R = quat_matrix(X[..., 3:])
    tR = skew(X[..., :3]) @ R
    Z = np.zeros_like(R)
    return np.concatenate([np.concatenate([R, tR], -1), np.concatenate([Z, R], -1)], -2)


def se3_Q(xi):
    """Barfoot Q(rho, phi) as PyPose calcQ."""
    xi = np.asarray(xi)
    tau, phi = xi[..., :3], xi[..., 3:]
    T, P = skew(tau), skew(phi)
    th = np.linalg.norm(phi, axis=-1)[..., None, None]
    big = th > _eps(xi)
    s = np.where(big, th, 1.0)
    s2 = s * s
    s4 = s2 * s2
    c1 = np.where(big, (s - np.sin(s)) / (s2 * s), 1.0 / 6.0)
    c2 = np.where(big, (s2 + 2 * np.cos(s) - 2) / (2 * s4), 1.0 / 24.0)
    c3 = np.where(big, (2 * s - 3 * np.sin(s) + s * np.cos(s)) / (2 * s4 * s), 1.0 / 120.0)
    PT, TP = P @ T, T @ P
    PTP = PT @ P
    return (0.5 * T + c1 * (PT + TP + PTP)
            + c2 * (P @ PT + TP @ P - 3 * PTP)
            + c3 * (PTP @ P + P @ PTP))


def se3_Jl_inv(xi):
    xi = np.asarray(xi)
    Ji = so3_Jl_inv(xi[..., 3:])
    Q = se3_Q(xi)
    Z = np.zeros_like(Ji)
    return np.concatenate([np.concatenate([Ji, -Ji @ Q @ Ji], -1),
                           np.concatenate([Z, Ji], -1)], -2)


def se3_Jl(xi):
    xi = np.asarray(xi)
    J = so3_Jl(xi[..., 3:])
    Q = se3_Q(xi)
    Z = np.zeros_like(J)
    return np.concatenate([np.concatenate([J, Q], -1), np.concatenate([Z, J], -1)], -2)


def from_matrix_SE3(M):
    """4x4 -> SE3 (used for the axis-permutation T of tartan2kitti)."""
    from scipy.spatial.transform import Rotation
    M = np.asarray(M, dtype=np.float64)
    q = Rotation.from_matrix(M[:3, :3]).as_quat()
    if q[3] < 0:
        q = -q
    return np.concatenate([M[:3, 3], q])
