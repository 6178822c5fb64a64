"""The bilevel ("imperative") training loop of the reference driver (reference train.py:162-308) as a reusable
harness on the islam_amd modules: per batch VO forward -> camera->IMU frame change -> pose chaining -> IMU
pre-integration (world + motion mode) -> PVGO -> one-step back-propagation; per epoch one optimizer step.

State carried between batches exactly like train.py: T0 = last PVGO pose (:219), init_state = last PVGO
pose/velocity with re-normalised quaternion (:297-299), gradients accumulated over the trajectory (:174-179),
the seven trajectory lists of init_epoch (:28-48) written by snapshot() in the formats of train.py:51-61, and the
alternating 'vo' / 'imu' epochs (:163, :207-212, :174-179): an IMU epoch reuses the previous epoch's VO motions
(no VO forward, no VO gradient) and steps the denoiser's optimizer.
"""
import os
import time

import numpy as np
import torch

from . import lietensor as pp
from .pvgo import run_pvgo
from .transformation import motion2pose_pypose, pose2motion_pypose


import os as _os
PREFETCH_FIRST = _os.environ.get('ISLAM_PREFETCH_FIRST', '1') == '1'

class BilevelLoop:
    def __init__(self, tartanvo, imu_module, rgb2imu_pose, imu_init, loss_weight=(1, 0.1, 10, 0.1), rot_w=1.0,
                 trans_w=0.1, lr=3e-6, batch_size=8, device='cuda:0', train_imu_denoiser=False):
        self.vo, self.imu, self.T_IL = tartanvo, imu_module, rgb2imu_pose
        self.loss_weight, self.rot_w, self.trans_w, self.bs, self.device = loss_weight, rot_w, trans_w, batch_size, device
        self.optimizer = torch.optim.Adam(tartanvo.vonet.flowPoseNet.parameters(), lr=lr)        # train.py:115-116
        self.imu_optimizer = None
        if getattr(imu_module, 'use_denoise_model', False):                                       # train.py:144-145
            self.imu_optimizer = torch.optim.Adam(imu_module.denoiser.parameters(), lr=3e-5)
        # False = the reference as released: the denoiser runs under eval=True, so IMU epochs step its optimizer on empty
        # gradients.  True = the fix of SURVEY section 8f rank 4: IMU epochs integrate differentiably
        # (IMUModule.train_denoiser) and the IMU-target loss of pvgo.py:95-111 reaches the denoiser's parameters.
        self.train_imu_denoiser = bool(train_imu_denoiser)
        self.prev_vo_motions = None                                                               # train.py:164
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
        self.pgo_motions, self.imu_motions = [], []
        self.current_idx = 0
        self.timing = dict(vo=0.0, imu=0.0, pgo=0.0, opt=0.0)

    def step(self, sample, target='vo', next_sample=None):
        """One pass of train.py:200-299 over one batch of ``batch_size`` frames.  ``next_sample``: the following batch (or a
        tuple of the following batches, nearest first); their frozen flow / disparity forwards are queued on a side stream
        (TartanVO.prefetch) -- one batch ahead: BEFORE this batch's pose head, whose host glue synchronises with the device twice
        (queued behind it the GPU sat idle for ~2 ms per batch: 511 -> ~600 frames/s) -- and overlap with this batch's pose head /
        IMU / PVGO / backward.
        With TWO batches ahead the side stream never runs dry: with one, nothing but the launch-bound pose head runs between
        the end of batch k+1's frozen forward and the start of batch k+2's."""
        bs, dev = self.bs, self.device
        # stage timing: the sequential schedule synchronises after every stage (self.timing = device-inclusive stage times); the
        # pipelined one must not -- every hand-over to the host (pose -> host glue, IMU / PVGO results) synchronises by itself, and a
        # barrier behind the backward would only stop the host from enqueueing the next batch while the GPU still runs this one's
        # (self.timing then holds host times per stage)
        sync = (lambda: None) if next_sample is not None else torch.cuda.synchronize
        t0 = time.perf_counter()
        motions = None
        if target != 'vo' and self.prev_vo_motions is not None:          # train.py:207-209: IMU epochs reuse last epoch's VO
            motions = self.prev_vo_motions[self.current_idx:self.current_idx + bs]
            if len(motions) != bs:                                        # (the reference's bare `except:` falls back to VO, Q15)
                motions = None
        if motions is None:
            # a VO forward inside an IMU epoch (first epoch, or a short slice of last epoch's motions) must not leave gradients
            # on the pose head: the next 'vo' epoch's optimizer.step() would apply them (the IMU epoch never zeroes them)
            ahead = [n for n in (next_sample if isinstance(next_sample, (tuple, list)) else (next_sample,)) if n is not None] \
                if next_sample is not None and hasattr(self.vo, 'prefetch') else []

            def prefetch_next(batches):
                for nxt in batches:
                    self.vo.prefetch(nxt)
            # the next batch's frozen forward is queued BEFORE this batch's pose head and host glue (ISLAM_PREFETCH_FIRST=0: behind them; the glue
            # synchronises with the device twice; queued behind it, the side stream sat idle for ~2 ms per batch) -- but no more batches
            # than there are captured graph copies: a replay on a copy whose previous replay is still running makes the HOST wait
            # (VONet._frozen_graphed's per-copy fence) before anything of this batch's main chain is enqueued; the rest goes behind the forward
            first = PREFETCH_FIRST
            n_first = len(ahead) if not first else min(len(ahead), max(1, int(getattr(getattr(self.vo, 'vonet', None), 'graph_instances', 1))))
            if first:
                prefetch_next(ahead[:n_first])
            # no autograd state for the VO forward of an IMU epoch (TartanVO.forward opens its own grad mode: an outer
            # torch.set_grad_enabled would not reach it)
            if self.__dict__.get('_vo_takes_need_grad') is None:
                import inspect
                fwd = getattr(self.vo, 'forward', self.vo)
                self._vo_takes_need_grad = 'need_grad' in inspect.signature(fwd).parameters
            if self._vo_takes_need_grad:
                res = self.vo(sample, need_grad=(target == 'vo'))
            else:                                                  # a VO front-end without the flag (stand-ins in tests)
                with torch.set_grad_enabled(target == 'vo'):
                    res = self.vo(sample)
            prefetch_next(ahead if not first else ahead[n_first:])
            motions = res.get('motion_host', res['motion'])      # TartanVO(host_glue=True): the same motions, on the host
            T_IL = self.T_IL.to(motions.device).to(motions.dtype)
            motions = T_IL @ motions @ T_IL.Inv()                                               # train.py:214-215
            if target != 'vo':
                motions = motions.detach()
        sync(); t1 = time.perf_counter()
        # VO-only dead reckoning is book-keeping (train.py:219-228 keeps it for the plots): no gradient flows through it,
        # so the 8 sequential SE3 products run on the host copy instead of ~160 tiny device launches
        motions_host = pp.SE3(motions.detach().tensor().cpu())
        with torch.no_grad():
            poses_vo = motion2pose_pypose(motions_host[:bs], torch.as_tensor(self.vo_poses[-1]).to(motions_host.dtype))
        self.vo_motions.extend(motions_host.tensor().numpy())
        self.vo_poses.extend(poses_vo.tensor().numpy()[1:])

        st, end = self.current_idx, self.current_idx + bs
        if hasattr(self.imu, 'train_denoiser'):
            self.imu.train_denoiser = self.train_imu_denoiser and target == 'imu' and self.imu_optimizer is not None
        # train.py:231-247 calls integrate twice on the same range (world rows, then motion rows); IMUModule.integrate_both produces
        # the pair from one pass over the samples (same values bit for bit), any other integrator object gets the two calls
        if hasattr(self.imu, 'integrate_both'):
            (imu_trans, imu_rots, _, imu_vels), (imu_dtrans, imu_drots, _, imu_dvels) = self.imu.integrate_both(st, end, self.init_state)
        else:
            imu_trans, imu_rots, _, imu_vels = self.imu.integrate(st, end, self.init_state, motion_mode=False)
            imu_dtrans, imu_drots, _, imu_dvels = self.imu.integrate(st, end, self.init_state, motion_mode=True)
        imu_poses = pp.SE3(torch.cat((imu_trans, imu_rots.tensor()), axis=1))
        self.imu_poses.extend(imu_poses[1:].detach().numpy())
        with torch.no_grad():
            self.imu_motions.extend(pose2motion_pypose(pp.SE3(imu_poses.detach().tensor())).numpy())    # train.py:240-242
        sync(); t2 = time.perf_counter()

        links = sample['link'] - self.current_idx
        trans_loss, rot_loss, pgo_poses, pgo_vels, _ = run_pvgo(imu_poses, imu_vels, motions, links, sample['dt'], imu_drots,
                                                                 imu_dtrans, imu_dvels, device=dev, radius=1e4,
                                                                 loss_weight=self.loss_weight, target=target)
        pgo_poses_np, pgo_vels_np = pgo_poses.numpy(), pgo_vels.numpy()
        self.pgo_motions.extend(pose2motion_pypose(pgo_poses).numpy())                          # train.py:264-266
        self.pgo_poses.extend(pgo_poses_np[1:])
        self.pgo_vels.extend(pgo_vels_np[1:])
        sync(); t3 = time.perf_counter()

        loss_bp = torch.cat((self.rot_w * rot_loss, self.trans_w * trans_loss))                # train.py:280
        # the value that is returned: read BEFORE the backward is enqueued -- a device-to-host read behind it would wait for the
        # whole backward pass, and the pipelined schedule wants the host back while the GPU still runs it
        tl_h, rl_h = getattr(trans_loss, 'host', None), getattr(rot_loss, 'host', None)
        if tl_h is not None and rl_h is not None:      # run_pvgo already brought the loss vectors to the host with the poses
            loss_value = float(self.rot_w * rl_h.sum() + self.trans_w * tl_h.sum())
        else:
            loss_value = float(loss_bp.detach().sum())
        if loss_bp.requires_grad:
            self._accumulate_gradients(loss_bp)
        sync(); t4 = time.perf_counter()

        self.current_idx += bs
        rot = pgo_poses_np[-1][3:].astype(np.float64)
        self.init_state = dict(rot=rot / np.linalg.norm(rot), pos=pgo_poses_np[-1][:3].astype(np.float64),
                               vel=pgo_vels_np[-1].astype(np.float64))
        for k, v in zip(('vo', 'imu', 'pgo', 'opt'), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            self.timing[k] += v
        return loss_value

    def _accumulate_gradients(self, loss_bp):
        """loss_bp.backward(ones) (train.py:282-283) for the parameters the two optimizers own: the gradients of this batch are
        taken with torch.autograd.grad and added to .grad with ONE multi-tensor add instead of one AccumulateGrad launch per
        parameter (the optimizer steps once per epoch, so every batch but the first accumulates: ~110 tiny launches per batch).
        Same arithmetic: grad <- grad + g in fp32."""
        params = [p for opt in (self.optimizer, self.imu_optimizer) if opt is not None
                  for grp in opt.param_groups for p in grp['params'] if p.requires_grad]
        if not params:                               # everything frozen: nothing to accumulate (autograd.grad rejects an empty list)
            return
        # a pose head on nets._PoseGraph accumulates its own gradients inside its backward node; its leaf is listed so that the engine
        # runs that node (its parameters then come back as unused)
        vonet = getattr(self.vo, 'vonet', None)
        leaf = vonet.pose_graph_leaf() if hasattr(vonet, 'pose_graph_leaf') else None
        grads = torch.autograd.grad(loss_bp, params + ([leaf] if leaf is not None else []), torch.ones_like(loss_bp), allow_unused=True)[:len(params)]
        acc, new = [], []
        for p, g in zip(params, grads):
            if g is None:
                continue
            if p.grad is None:
                p.grad = g.detach().clone()          # (a graphed backward hands out static buffers: never keep them)
            else:
                acc.append(p.grad)
                new.append(g.detach())
        if acc:
            torch._foreach_add_(acc, new)

    def snapshot(self, trainroot, epoch):
        """train.py:51-61: the seven text files of an epoch directory (the de-facto output contract)."""
        d = os.path.join(trainroot, str(epoch))
        os.makedirs(d, exist_ok=True)
        for name, rows in (('vo_pose', self.vo_poses), ('vo_motion', self.vo_motions), ('pgo_pose', self.pgo_poses),
                           ('pgo_motion', self.pgo_motions), ('pgo_vel', self.pgo_vels), ('imu_pose', self.imu_poses),
                           ('imu_motion', self.imu_motions)):
            np.savetxt(os.path.join(d, name + '.txt'), np.stack(rows))
        return d

    def end_epoch(self, target='vo', trainroot=None, epoch=1, reset=False):
        """train.py:172-198: one optimizer step per pass over the trajectory (the VO head in 'vo' epochs, the denoiser in
        'imu' epochs), final snapshot, the epoch's VO motions kept for the next (IMU) epoch, lists re-initialised."""
        if target == 'vo':
            self.optimizer.step()
            self.optimizer.zero_grad()
        elif target == 'imu' and self.imu_optimizer is not None:
            self.imu_optimizer.step()
            self.imu_optimizer.zero_grad()
        if trainroot is not None:
            self.snapshot(trainroot, epoch)
        if self.vo_motions:
            self.prev_vo_motions = pp.SE3(torch.as_tensor(np.stack(self.vo_motions)).to(self.device))       # train.py:193
        if reset:
            self.reset()
