"""Seeded synthetic inputs of the shapes BASELINE.json's configs name.

No datasets or pretrained weights exist in the build container or on the GPU box, so every
benchmark and parity test runs on these generators.  Pure numpy/scipy host code; nothing
here is on the timed path.
"""
import numpy as np
from scipy.spatial.transform import Rotation


def _so3_exp_q(phi):
    return Rotation.from_rotvec(phi).as_quat()   # [x,y,z,w]


def car_trajectory(n_frames, frame_dt=0.1, imu_per_frame=10, seed=11, gravity=9.81,
                   gyro_sigma=1.7e-4, acc_sigma=2e-3, vo_sigma_t=0.05, vo_sigma_r=0.002, vo_seed=7):
    """Config-4 'KITTI-shape' planar-car ground truth, 100 Hz IMU and noisy VO motions.

    Body frame: x forward, z up; world gravity vector [0,0,g] as pp.module.IMUPreintegrator
    assumes (a_world = R*acc - g).  Returns a dict of float64 arrays:
      gt_pos (F,3), gt_quat (F,4) xyzw, gt_vel (F,3) world frame, at the F=n_frames camera times
      accels, gyros (S,3) body frame, imu_dts (S,), rgb2imu_sync (F,) int64, S = (F-1)*imu_per_frame + 1
      vo_motions (F-1,7) relative SE3 motions X_k^-1 X_{k+1} with noise, links (F-1,2), dts (F-1,)
    """
    F = int(n_frames)
    S = (F - 1) * imu_per_frame + 1
    h = frame_dt / imu_per_frame
    sub = 8                                     # GT integration sub-steps per IMU sample
    hs = h / sub
    T = np.arange((S + 1) * sub + 1) * hs

    speed = 10.0 + 2.0 * np.sin(0.05 * T)
    yaw_rate = 0.1 * np.sin(0.02 * T)
    yaw = np.concatenate([[0.0], np.cumsum(0.5 * (yaw_rate[1:] + yaw_rate[:-1]) * hs)])
    roll = 0.01 * np.sin(0.3 * T)
    pitch = 0.01 * np.sin(0.2 * T)
    R = Rotation.from_euler('ZYX', np.stack([yaw, pitch, roll], 1))
    vel_w = R.apply(np.stack([speed, np.zeros_like(speed), np.zeros_like(speed)], 1))
    pos_w = np.concatenate([np.zeros((1, 3)), np.cumsum(0.5 * (vel_w[1:] + vel_w[:-1]) * hs, 0)])

    idx = np.arange(S) * sub                    # IMU sample instants
    # body rate over the NEXT imu interval, specific force at the interval start
    Rn = R[idx + sub]
    Rc = R[idx]
    gyro = (Rc.inv() * Rn).as_rotvec() / h
    acc_w = (vel_w[idx + sub] - vel_w[idx]) / h
    acc = Rc.inv().apply(acc_w + np.array([0.0, 0.0, gravity]))
    rng = np.random.default_rng(seed)
    gyro = gyro + rng.normal(0.0, gyro_sigma, gyro.shape)
    acc = acc + rng.normal(0.0, acc_sigma, acc.shape)

    fidx = np.arange(F) * imu_per_frame
    gt_R = R[fidx * sub]
    gt_quat = gt_R.as_quat()
    gt_quat = np.where(gt_quat[:, 3:] < 0, -gt_quat, gt_quat)
    gt_pos = pos_w[fidx * sub]
    gt_vel = vel_w[fidx * sub]

    # relative motions in the body frame + noise: m_k = X_k^-1 X_{k+1} * Exp(noise)
    rel_R = gt_R[:-1].inv() * gt_R[1:]
    rel_t = gt_R[:-1].inv().apply(gt_pos[1:] - gt_pos[:-1])
    vrng = np.random.default_rng(vo_seed)
    n_t = vrng.normal(0.0, vo_sigma_t, rel_t.shape)
    n_r = vrng.normal(0.0, vo_sigma_r, rel_t.shape)
    vo_R = rel_R * Rotation.from_rotvec(n_r)
    vo_t = rel_t + rel_R.apply(n_t)
    vo_q = vo_R.as_quat()
    vo_q = np.where(vo_q[:, 3:] < 0, -vo_q, vo_q)
    links = np.stack([np.arange(F - 1), np.arange(1, F)], 1).astype(np.int64)
    return dict(gt_pos=gt_pos, gt_quat=gt_quat, gt_vel=gt_vel,
                accels=acc, gyros=gyro, imu_dts=np.full(S, h), rgb2imu_sync=fidx.astype(np.int64),
                vo_motions=np.concatenate([vo_t, vo_q], 1), links=links, dts=np.full(F - 1, frame_dt),
                gravity=gravity,
                init=dict(pos=gt_pos[0].copy(), rot=gt_quat[0].copy(), vel=gt_vel[0].copy()))


def pvgo_problem_from_deltas(traj, drots, dtrans, dvels, imu_pos, imu_rot, imu_vel):
    """Pack run_pvgo's positional inputs from a trajectory dict and integrated IMU quantities."""
    init_nodes = np.concatenate([imu_pos, imu_rot], 1)
    return dict(init_nodes=init_nodes, init_vels=imu_vel, vo_motions=traj['vo_motions'], links=traj['links'],
                dts=traj['dts'], imu_drots=drots, imu_dtrans=dtrans, imu_dvels=dvels)


def stereo_batch(batch, height=448, width=640, seed=1234, max_flow=8.0, disp_range=(5.0, 40.0)):
    """Synthetic sample dict of the shape TartanVO consumes (SURVEY section 8b sample-dict contract).

    Band-limited random texture; img1 = img0 translated by a constant flow per sample, right image =
    left shifted by a constant disparity per sample.  float32 torch tensors, batch-first.
    """
    import torch
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand(batch, 3, height // 8 + 12, width // 8 + 24, generator=g)      # 48 / 96 px margins
    base = torch.nn.functional.interpolate(lo, scale_factor=8, mode='bicubic', align_corners=False).clamp(0, 1)
    flows = (torch.rand(batch, 2, generator=g) * 2 - 1) * max_flow * 4      # px at full res
    disps = disp_range[0] + torch.rand(batch, generator=g) * (disp_range[1] - disp_range[0])

    def crop(img, dx, dy):
        x0 = 96 + int(round(dx))
        y0 = 48 + int(round(dy))
        return img[:, y0:y0 + height, x0:x0 + width]

    img0 = torch.stack([crop(base[b], 0, 0) for b in range(batch)])
    img1 = torch.stack([crop(base[b], -flows[b, 0].item(), -flows[b, 1].item()) for b in range(batch)])
    img0_r = torch.stack([crop(base[b], disps[b].item(), 0) for b in range(batch)])
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    fx = fy = 718.856 * 448 / 375
    cx, cy = 607.1928 * 448 / 375 - 422.0 * 0 - 85.0, 185.2157 * 448 / 375
    ww, hh = torch.meshgrid(torch.arange(width, dtype=torch.float32), torch.arange(height, dtype=torch.float32),
                            indexing='xy')
    layer = torch.stack([(ww + 0.5 - cx) / fx, (hh + 0.5 - cy) / fy])[:, ::4, ::4]
    return {
        'img0': img0.contiguous(), 'img1': img1.contiguous(), 'img0_r': img0_r.contiguous(),
        'img0_norm': ((img0 - mean) / std).contiguous(), 'img0_r_norm': ((img0_r - mean) / std).contiguous(),
        'intrinsic': layer.unsqueeze(0).repeat(batch, 1, 1, 1).contiguous(),
        'intrinsic_calib': torch.tensor([[fx, fy, cx, cy]], dtype=torch.float32).repeat(batch, 1),
        'extrinsic': torch.tensor([[0.54, 0, 0, 0, 0, 0, 1.0]], dtype=torch.float32).repeat(batch, 1),
        'datatype': ['kitti'] * batch,
        'link': torch.stack([torch.arange(batch), torch.arange(1, batch + 1)], 1),
        'dt': torch.full((batch,), 0.1),
    }
