"""The bilevel ("imperative") training loop of the reference driver (reference train.py:162-308) as a reusable
harness on the islam_amd modules: per batch VO forward -> camera->IMU frame change -> pose chaining -> IMU
pre-integration (world + motion mode) -> PVGO -> one-step back-propagation; per epoch one optimizer step.

State carried between batches exactly like train.py: T0 = last PVGO pose (:219), init_state = last PVGO
pose/velocity with re-normalised quaternion (:297-299), gradients accumulated over the trajectory (:174-179).
"""
import time

import numpy as np
import torch

from . import lietensor as pp
from .pvgo import run_pvgo
from .transformation import motion2pose_pypose, pose2motion_pypose


class BilevelLoop:
    def __init__(self, tartanvo, imu_module, rgb2imu_pose, imu_init, loss_weight=(1, 0.1, 10, 0.1), rot_w=1.0,
                 trans_w=0.1, lr=3e-6, batch_size=8, device='cuda:0'):
        self.vo, self.imu, self.T_IL = tartanvo, imu_module, rgb2imu_pose
        self.loss_weight, self.rot_w, self.trans_w, self.bs, self.device = loss_weight, rot_w, trans_w, batch_size, device
        self.optimizer = torch.optim.Adam(tartanvo.vonet.flowPoseNet.parameters(), lr=lr)        # train.py:115-116
        self.imu_init = imu_init
        self.reset()

    def reset(self):
        """init_epoch (train.py:28-48)."""
        i = self.imu_init
        self.init_state = dict(rot=np.asarray(i['rot'], dtype=np.float64), pos=np.asarray(i['pos'], dtype=np.float64),
                               vel=np.asarray(i['vel'], dtype=np.float64))
        first = np.concatenate([i['pos'], i['rot']]).astype(np.float32)
        self.pgo_poses, self.vo_poses = [first], [first]
        self.vo_motions, self.pgo_vels, self.imu_poses = [], [np.asarray(i['vel'], dtype=np.float32)], [first]
        self.current_idx = 0
        self.timing = dict(vo=0.0, imu=0.0, pgo=0.0, opt=0.0)

    def step(self, sample, target='vo', next_sample=None):
        """One pass of train.py:200-299 over one batch of ``batch_size`` frames.  ``next_sample``: the following batch;
        its frozen flow / disparity forward is started on a side stream as soon as this batch's pose head is enqueued
        (TartanVO.prefetch) and overlaps with this batch's IMU / PVGO / backward."""
        bs, dev = self.bs, self.device
        sync = (lambda: torch.cuda.current_stream().synchronize()) if next_sample is not None else torch.cuda.synchronize
        t0 = time.perf_counter()
        res = self.vo(sample)
        if next_sample is not None and hasattr(self.vo, 'prefetch'):
            self.vo.prefetch(next_sample)
        motions = res.get('motion_host', res['motion'])      # TartanVO(host_glue=True): the same motions, on the host
        T_IL = self.T_IL.to(motions.device).to(motions.dtype)
        motions = T_IL @ motions @ T_IL.Inv()                                                   # train.py:214-215
        sync(); t1 = time.perf_counter()
        # VO-only dead reckoning is book-keeping (train.py:219-228 keeps it for the plots): no gradient flows through it,
        # so the 8 sequential SE3 products run on the host copy instead of ~160 tiny device launches
        motions_host = pp.SE3(motions.detach().tensor().cpu())
        with torch.no_grad():
            poses_vo = motion2pose_pypose(motions_host[:bs], torch.as_tensor(self.vo_poses[-1]).to(motions_host.dtype))
        self.vo_motions.extend(motions_host.tensor().numpy())
        self.vo_poses.extend(poses_vo.tensor().numpy()[1:])

        st, end = self.current_idx, self.current_idx + bs
        imu_trans, imu_rots, _, imu_vels = self.imu.integrate(st, end, self.init_state, motion_mode=False)
        imu_poses = pp.SE3(torch.cat((imu_trans, imu_rots.tensor()), axis=1))
        self.imu_poses.extend(imu_poses[1:].numpy())
        imu_dtrans, imu_drots, _, imu_dvels = self.imu.integrate(st, end, self.init_state, motion_mode=True)
        sync(); t2 = time.perf_counter()

        links = sample['link'] - self.current_idx
        trans_loss, rot_loss, pgo_poses, pgo_vels, _ = run_pvgo(imu_poses, imu_vels, motions, links, sample['dt'], imu_drots,
                                                                 imu_dtrans, imu_dvels, device=dev, radius=1e4,
                                                                 loss_weight=self.loss_weight, target=target)
        pgo_poses_np, pgo_vels_np = pgo_poses.numpy(), pgo_vels.numpy()
        self.pgo_poses.extend(pgo_poses_np[1:])
        self.pgo_vels.extend(pgo_vels_np[1:])
        sync(); t3 = time.perf_counter()

        loss_bp = torch.cat((self.rot_w * rot_loss, self.trans_w * trans_loss))                # train.py:280
        if loss_bp.requires_grad:
            loss_bp.backward(torch.ones_like(loss_bp))
        sync(); t4 = time.perf_counter()

        self.current_idx += bs
        rot = pgo_poses_np[-1][3:].astype(np.float64)
        self.init_state = dict(rot=rot / np.linalg.norm(rot), pos=pgo_poses_np[-1][:3].astype(np.float64),
                               vel=pgo_vels_np[-1].astype(np.float64))
        for k, v in zip(('vo', 'imu', 'pgo', 'opt'), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            self.timing[k] += v
        return float(loss_bp.detach().sum())

    def end_epoch(self):
        """train.py:172-198: one optimizer step per pass over the trajectory."""
        self.optimizer.step()
        self.optimizer.zero_grad()
