"""The (B,6) -> (B,7) pose algebra behind the networks as ONE autograd node (TartanVO.py:107-198, dense_ba.py:88-176 for the scale).

TartanVO.forward(host_glue=True) used to run this part as ~100 LieTensor / torch operations on (8,.) host tensors -- unit rescale, so3 Exp,
the NED->camera conjugation (Datasets/transformation.py:89-98), the stereo scale with its gradient re-attached through
T^-1.translation() / T^-1.rotation() (islam_amd/TartanVO.py::stereo_scale), normalisation, the final conjugation -- and autograd walked
as many nodes back.  On the pipelined step's main chain that is ~2.4 ms of Python per batch (scripts/vio_hostprof.py), more on a slower
host.  Here the same arithmetic is written once in vectorised numpy with the backward in closed form, following the LieTensor shim's
(= PyPose's) gradient conventions node for node:
    SE3 product Z = X Y:   g_X = g[:6],  g_Y = g[:6] Ad(X)            (left-perturbation tangents, lietensor._SE3Mul)
    SE3 inverse Y = X^-1:  g_X = -g[:6] Ad(Y)                         (lietensor._SE3Inv)
    so3 Exp:               g_phi = g[:3] Jl(phi)                      (lietensor._so3Exp)
    SO3 action o = R p:    g_R = -g [o]x                              (lietensor._SO3Act)
    .translation() / .rotation() / SE3(cat(t, q)): raw slices of the 7-vector
tests/test_frontend_gpu.py::test_fused_pose_glue_matches_the_operator_by_operator_path holds values and gradients against the old path."""
import numpy as np
import torch

from . import lietensor as pp
from . import ops

_T_AXES_Q = None


def _axes():
    """(T, T^-1) of the NED -> camera axis permutation as float64 7-vectors (transformation._axes_pose)."""
    global _T_AXES_Q
    if _T_AXES_Q is None:
        from .transformation import _axes_pose
        T, Ti = _axes_pose(torch.float64, 'cpu')
        _T_AXES_Q = (T.tensor().numpy().copy(), Ti.tensor().numpy().copy())
    return _T_AXES_Q


def _mul(x, y):
    return np.concatenate([x[..., :3] + pp._qact_np(x[..., 3:], y[..., :3]), pp._qmul_np(x[..., 3:], y[..., 3:])], -1)


def _inv(x):
    qi = np.concatenate([-x[..., 3:6], x[..., 6:7]], -1)
    return np.concatenate([-pp._qact_np(qi, x[..., :3]), qi], -1)


def _adj(x):
    R = pp._qmat_np(x[..., 3:])
    top = np.concatenate([R, pp._skew_np(x[..., :3]) @ R], -1)
    return np.concatenate([top, np.concatenate([np.zeros_like(R), R], -1)], -2)


def _so3_exp(phi):
    """lietensor._so3_exp in numpy (same branches, same Taylor coefficients)."""
    th = np.linalg.norm(phi, axis=-1, keepdims=True)
    th2 = th * th
    th4 = th2 * th2
    big = th > np.finfo(phi.dtype).eps
    ths = np.where(big, th, 1.0)
    imag = np.where(big, np.sin(0.5 * ths) / ths, 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4)
    real = np.where(big, np.cos(0.5 * ths), 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4)
    return np.concatenate([phi * imag, real], -1)


def _so3_Jl(phi):
    """lietensor._so3_Jl in numpy."""
    K = pp._skew_np(phi)
    th = np.linalg.norm(phi, axis=-1)[..., None, None]
    big = th > np.finfo(phi.dtype).eps
    ths = np.where(big, th, 1.0)
    c1 = np.where(big, (1 - np.cos(ths)) / ths ** 2, 0.5)
    c2 = np.where(big, (ths - np.sin(ths)) / ths ** 3, 1.0 / 6.0)
    return np.eye(3, dtype=phi.dtype) + c1 * K + c2 * (K @ K)


def _normalize(x):
    """torch.nn.functional.normalize(x, dim=-1) and the norm it divided by."""
    n = np.maximum(np.linalg.norm(x, axis=-1, keepdims=True), 1e-12)
    return x / n, n


def _normalize_bwd(g, y, n):
    return (g - y * (y * g).sum(-1, keepdims=True)) / n


def _row(g, M):
    return np.einsum('...i,...ij->...j', g, M)


def glue_forward_np(p, use_kitti_coord, scale_fn):
    """The forward in numpy.  p: (B,6) float64 = the pose head's output x pose_std; scale_fn(pose_enu (B,7)) -> sums (B,18) float64 of
    islam_scale_ls at that pose (the one device call of the algebra; tests pass a stand-in).  Returns (motion (B,7), saved)."""
    tau, phi = p[:, :3], p[:, 3:]
    q = _so3_exp(phi)
    T, Ti = _axes()
    X0 = np.concatenate([tau, q], -1)
    pose_enu = _mul(_mul(T, X0), Ti)                                           # tartan2kitti_pypose(pose): the scale is recovered in this frame
    sums = scale_fn(pose_enu)
    sv = np.float32(1.0 / sums[:, 0] * sums[:, 1]).astype(np.float64)          # = the kernel's scale (scale_final_kernel's arithmetic)
    n, ntau = _normalize(tau)
    X = np.concatenate([n * sv[:, None], q], -1)
    M = _mul(_mul(T, X), Ti) if use_kitti_coord else X
    return M, (tau, phi, n, ntau, sv, pose_enu, sums)


def glue_backward_np(g6, saved, intr4, use_kitti_coord):
    """Gradient w.r.t. p (B,6) given the left-tangent gradient g6 (B,6) of the motion; conventions in the module docstring."""
    tau, phi, n, ntau, sv, pose_enu, sums = saved
    T, Ti = _axes()
    AdT = _adj(T)
    gX = _row(g6, AdT) if use_kitti_coord else g6                             # M = (T X) T^-1
    Jl = _so3_Jl(phi)
    g_trans = gX[:, :3]
    g_phi = _row(gX[:, 3:6], Jl)
    g_s = (n * g_trans).sum(-1)                                               # trans = n * s
    g_tau = _normalize_bwd(g_trans * sv[:, None], n, ntau)
    # ---- the scale's own dependence on the pose (TartanVO.stereo_scale): s = Mw / MM with M linear in a = K t^ and w linear in R
    fx, fy, cx, cy = intr4[:, 0], intr4[:, 1], intr4[:, 2], intr4[:, 3]
    MM = sums[:, 0]
    dMw_da = np.stack([-sums[:, 2], -sums[:, 3], sums[:, 4]], -1)
    dMM_da = np.stack([-2 * sums[:, 5], -2 * sums[:, 6], 2 * sums[:, 7]], -1)
    ga = (dMw_da - sv[:, None] * dMM_da) / MM[:, None]
    GR = np.stack([fx[:, None] * sums[:, 8:11], fy[:, None] * sums[:, 11:14], sums[:, 14:17]], 1) / MM[:, None, None]
    Tinv = _inv(pose_enu)
    tn, nt = _normalize(Tinv[:, :3])
    g_a = g_s[:, None] * ga
    g_tn = np.stack([fx * g_a[:, 0], fy * g_a[:, 1], cx * g_a[:, 0] + cy * g_a[:, 1] + g_a[:, 2]], -1)
    g_t = _normalize_bwd(g_tn, tn, nt)
    R = pp._qmat_np(Tinv[:, 3:])                                               # column j = R e_j
    g_cols = g_s[:, None, None] * GR
    gR = np.zeros_like(g_t)
    for j in range(3):
        gR -= _row(g_cols[:, :, j], pp._skew_np(R[:, :, j]))
    g_enu = -_row(np.concatenate([g_t, gR], -1), _adj(Tinv))                  # Tinv = pose_enu^-1
    gX0 = _row(g_enu, AdT)                                                    # pose_enu = (T X0) T^-1
    g_tau = g_tau + gX0[:, :3]
    g_phi = g_phi + _row(gX0[:, 3:6], Jl)
    return np.concatenate([g_tau, g_phi], -1)


class _FusedPoseGlue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose, std, disp, flow, intr4, baseline, edge, th, use_kitti_coord):
        dev = pose.device
        p = pose.detach().double().cpu().numpy() * std                       # (B,6), the one read of the network output
        out = {}

        def scale_fn(pose_enu):
            out['s'], out['z'], out['mask'], out['dmask'], sums = ops.scale_ls(disp, flow, torch.from_numpy(pose_enu).to(dev), intr4, baseline, edge, th)
            return sums.cpu().numpy()                                          # (B,18) float64: the second (and last) read
        M, saved = glue_forward_np(p, use_kitti_coord, scale_fn)
        ctx.saved = (std, saved, intr4.detach().double().cpu().numpy(), bool(use_kitti_coord), dev, pose.dtype)
        ctx.mark_non_differentiable(out['s'], out['z'], out['mask'], out['dmask'])
        return torch.from_numpy(M), out['s'], out['z'], out['mask'], out['dmask']

    @staticmethod
    def backward(ctx, g, *_unused):
        std, saved, intr4, kitti, dev, dtype = ctx.saved
        gp = glue_backward_np(g.detach().double().cpu().numpy()[:, :6], saved, intr4, kitti) * std
        return (torch.from_numpy(gp).to(dev, dtype),) + (None,) * 8


def fused_pose_glue(pose, pose_std, disp, flow, intr4, baseline, edge, th, use_kitti_coord):
    """pose: the pose head's (B,6) output on the device.  Returns (motion: SE3 LieTensor (B,7) float64 on the HOST, differentiable
    w.r.t. ``pose``; scale, depth, mask, depth_mask on the device)."""
    std = np.asarray(pose_std, dtype=np.float64)
    M, s, z, mask, dmask = _FusedPoseGlue.apply(pose, std, disp, flow, intr4, baseline, edge, th, use_kitti_coord)
    return pp.SE3(M), s, z, mask, dmask
