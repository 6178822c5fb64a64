"""Oracle (test infrastructure): the glue of TartanVO.forward after the networks (reference TartanVO.py:107-198,
`correct_scale=False` branch = stereo scale recovery), in numpy float64 on float32 network outputs.

    pose *= pose_std (:108); flow *= 5 (:122); disp *= 50/4 (:126); pose_ENU = tartan2kitti(pose) (:142)
    edge = Canny mask (:145-155, oracle/canny.py); per sample scale_from_disp_flow (:159-167, oracle/scale.py)
    trans = normalize(pose[:, :3]) * scale (:182); motion = tartan2kitti(pose) | cvtSE3(pose) (:194-197)
Frame change: Datasets/transformation.py:72-98 (cvtSE3_pypose, tartan2kitti_pypose)."""
import numpy as np

from . import canny, lie, scale

POSE_STD = np.array([0.13, 0.13, 0.13, 0.013, 0.013, 0.013])           # TartanVO.py:32
T_AXES = np.array([[0., 1, 0, 0], [0, 0, 1, 0], [1, 0, 0, 0], [0, 0, 0, 1]])      # transformation.py:91-94
DISP_TH = {'kitti': 5, 'tartanair': 1, 'euroc': 1}                      # TartanVO.py:161


def cvt_se3(pose6):
    pose6 = np.asarray(pose6, np.float64)
    return np.concatenate([pose6[..., :3], lie.so3_exp(pose6[..., 3:])], -1)


def tartan2kitti(pose6):
    T = lie.from_matrix_SE3(T_AXES)
    return lie.se3_mul(lie.se3_mul(T[None], cvt_se3(pose6)), lie.se3_inv(T)[None])


def forward_glue(flow, disp, pose, img0, intrinsic_calib, baseline, datatype, use_kitti_coord=True, edge=None):
    """flow (B,2,h,w), disp (B,1,h,w), pose (B,6): raw network outputs.  Returns the dict TartanVO.forward returns."""
    pose = np.asarray(pose, np.float64) * POSE_STD
    flow = np.asarray(flow, np.float32) * np.float32(5)
    disp = np.asarray(disp, np.float32) * np.float32(50 / 4)
    enu = tartan2kitti(pose)
    if edge is None:
        edge = canny.edge_mask(img0)
    res = dict(flow=flow, disp=disp, edge=edge, scale=[], mask=[], depth=[], depth_mask=[])
    for i in range(pose.shape[0]):
        fx, fy, cx, cy = np.asarray(intrinsic_calib[i], np.float32) / np.float32(4)
        s, z, m, dm, _ = scale.scale_from_disp_flow(disp[i], flow[i], enu[i], fx, fy, cx, cy, baseline[i], edge_mask=edge[i],
                                                    disp_th=DISP_TH[datatype[i]])
        for k, v in (('scale', s), ('mask', m), ('depth', z), ('depth_mask', dm)):
            res[k].append(v)
    for k in ('scale', 'mask', 'depth', 'depth_mask'):
        res[k] = np.stack(res[k])
    t = pose[:, :3]
    trans = t / np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-12) * res['scale'].astype(np.float64)[:, None]
    pose = np.concatenate([trans, pose[:, 3:]], 1)
    res['motion'] = tartan2kitti(pose) if use_kitti_coord else cvt_se3(pose)
    return res
