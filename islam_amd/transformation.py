"""The PyPose-side helpers of the reference's Datasets/transformation.py:72-124, on the LieTensor shim."""
import torch

from . import lietensor as pp


def cvtSE3_pypose(motion):
    """transformation.py:72-87: (t, so3) 6-vectors / 7-vectors / se3 / SE3 -> SE3."""
    if isinstance(motion, pp.LieTensor):
        if motion.ltype is pp.SE3_type:
            return motion.clone()
        if motion.ltype is pp.se3_type:
            return motion.Exp()
    else:
        if not isinstance(motion, torch.Tensor):
            motion = torch.tensor(motion)
        if motion.shape[-1] == 6:
            rot = pp.so3(motion[..., 3:]).Exp().tensor()
            return pp.SE3(torch.cat([motion[..., :3], rot], dim=-1))
        if motion.shape[-1] == 7:
            return pp.SE3(motion)
    assert False, "Not valid input."


_T_AXES = [[0., 1., 0., 0.], [0., 0., 1., 0.], [1., 0., 0., 0.], [0., 0., 0., 1.]]


_T_CACHE = {}


def _axes_pose(dtype, device):
    """(T, T^-1) of the NED->camera axis permutation as SE3, built once per (dtype, device) (transformation.py:91-97 rebuilds it per call)."""
    key = (dtype, str(device))
    hit = _T_CACHE.get(key)
    if hit is None:
        with torch.no_grad():
            T = pp.from_matrix(torch.tensor(_T_AXES, dtype=dtype), ltype=pp.SE3_type).to(device)
            hit = _T_CACHE[key] = (T, T.Inv())
    return hit


def tartan2kitti_pypose(motion):
    """transformation.py:89-98: conjugation by the NED->camera axis permutation."""
    motion = cvtSE3_pypose(motion)
    T, Tinv = _axes_pose(motion.dtype, motion.device)
    return T @ motion @ Tinv


def _host_nograd(t):
    return (not t.is_cuda) and t.dtype in (torch.float32, torch.float64) and not (torch.is_grad_enabled() and t.requires_grad)


def motion2pose_pypose(motion, T=None):
    """transformation.py:100-114: sequential prefix product T_k = T_{k-1} * m_k (order kept: it fixes rounding)."""
    motion = cvtSE3_pypose(motion)
    if _host_nograd(motion) and (T is None or (isinstance(T, torch.Tensor) and _host_nograd(cvtSE3_pypose(T)))):
        # book-keeping on the host (BilevelLoop's VO dead reckoning): the same products in the same order on numpy views
        m = pp._np(motion.tensor())
        out = pp.np.empty((m.shape[0] + 1, 7), dtype=m.dtype)
        out[0] = (0, 0, 0, 0, 0, 0, 1) if T is None else pp._np(cvtSE3_pypose(T).tensor()).reshape(7).astype(m.dtype)
        for k in range(m.shape[0]):
            out[k + 1, :3] = out[k, :3] + pp._qact_np(out[k, 3:], m[k, :3])
            out[k + 1, 3:] = pp._qmul_np(out[k, 3:], m[k, 3:])
        return pp.SE3(torch.from_numpy(out))
    if T is None:
        T = pp.SE3(torch.tensor([0, 0, 0, 0, 0, 0, 1], dtype=motion.dtype)).to(motion.device)
    else:
        T = cvtSE3_pypose(T).to(motion.device)
    pose = [T]
    for k in range(motion.shape[0]):
        T = T @ motion[k]
        pose.append(T)
    return pp.SE3(torch.stack([p.tensor() for p in pose]))


def pose2motion_pypose(pose):
    """transformation.py:116-124: m_i = T_i^-1 * T_{i+1} (batched: the products are independent)."""
    pose = cvtSE3_pypose(pose)
    if _host_nograd(pose):
        p = pp._np(pose.tensor())
        a, b = p[:-1], p[1:]
        qi = pp.np.concatenate([-a[:, 3:6], a[:, 6:7]], -1)                      # T_i^-1 = (-R^T t, R^T)
        ti = -pp._qact_np(qi, a[:, :3])
        return pp.SE3(torch.from_numpy(pp.np.concatenate([ti + pp._qact_np(qi, b[:, :3]), pp._qmul_np(qi, b[:, 3:])], -1)))
    return pose[:-1].Inv() @ pose[1:]
