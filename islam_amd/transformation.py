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


def tartan2kitti_pypose(motion):
    """transformation.py:89-98: conjugation by the NED->camera axis permutation."""
    motion = cvtSE3_pypose(motion)
    T = pp.from_matrix(torch.tensor(_T_AXES, dtype=motion.dtype), ltype=pp.SE3_type).to(motion.device)
    return T @ motion @ T.Inv()


def motion2pose_pypose(motion, T=None):
    """transformation.py:100-114: sequential prefix product T_k = T_{k-1} * m_k (order kept: it fixes rounding)."""
    motion = cvtSE3_pypose(motion)
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
    return pose[:-1].Inv() @ pose[1:]
