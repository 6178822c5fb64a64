"""BASELINE configs[0] at FULL length with the REAL nets: a 271-frame KITTI-shape sequence = 270 links
= 33 bilevel steps of B = 8 (drop_last, train.py:95-97) + the epoch end (one Adam step on the pose head, the seven snapshot files,
train.py:172-198) -- the benched configuration (bf16 execution copies of the frozen nets, both HIP graphs, the prefetch schedule, host
glue, channels-last pose head) against the fp32 eager networks driven through the ORACLE's glue, IMU pre-integration and PVGO
(oracle/tartanvo.py, oracle/imu.py, oracle/pvgo.py), window by window with the same state hand-over (train.py:219, :297-299).

What bf16 operands in the frozen nets may cost over a whole trajectory is bounded here, not only per window
(tests/test_benched_frontend_gpu.py): the absolute trajectory error of the PGO trajectory against the synthetic ground truth must be
the same within 1 % (north_star: "ATE within 1 % of reference"), and the two PGO trajectories must stay together over all 264 frames.
Reference: train.py:162-308."""
import os

import numpy as np
import pytest
import torch

from islam_amd import evaluate, synthetic
from oracle import imu as oimu
from oracle import lie
from oracle import pvgo as opvgo
from oracle import tartanvo as otv

pytestmark = pytest.mark.gpu
B, FRAMES = 8, 271
STEPS = (FRAMES - 1) // B          # 33
LW = (1, 0.1, 10, 0.1)             # run_kitti.sh:5
# Bounds over the WHOLE sequence.  Per window the bf16 nets move a VO translation by <= 8e-2 of its norm and a rotation by <= 3e-3 rad
# (tests/test_benched_frontend_gpu.py); PVGO ties every window to the IMU pre-integration (weights 0.1 / 10 / 0.1 against 1 for VO),
# and the next window starts from the optimised state: the difference between the two pipelines grows along the path, slowly
# (the IMU factors re-anchor every window).  Measured on MI355X (2026-10, random weights): ATE 82.447 vs 82.499 m
# (6.3e-4 relative; both pipelines are equally far from the ground truth because random-weight nets predict noise -- what is
# compared is the two pipelines), trajectories apart by at most 0.18 m = 6.1e-4 of the 294 m path and 8.7e-7 rad, window losses within 1e-5.
TOL_ATE_REL = 1e-2                 # |ATE_bf16 - ATE_ref| / ATE_ref  (north_star: "ATE within 1 % of reference")
TOL_DRIFT_POS = 3e-3               # max_k |p_bf16(k) - p_ref(k)| / path length  (5x the measured value)
TOL_DRIFT_ROT = 1e-4               # max_k angle(R_ref(k)^-1 R_bf16(k)), rad


def _make(**kw):
    from islam_amd.TartanVO import TartanVO
    torch.manual_seed(0)
    vo = TartanVO(correct_scale=False, fix_parts=('flow', 'stereo'), use_kitti_coord=True, **kw)
    with torch.no_grad():      # random weights predict garbage disparity: pin the stereo head to 10 px (as bench.py does)
        vo.vonet.stereoNet.conv_c13.weight.zero_()
        vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
    return vo


def test_whole_kitti04_length_epoch_with_the_real_nets(cuda, tmp_path):
    from islam_amd import lietensor as pp
    from islam_amd.bilevel import BilevelLoop
    from islam_amd.edges import edge_mask
    from islam_amd.imu_integrator import IMUModule
    tr = synthetic.car_trajectory(FRAMES, seed=3)
    base_batches = []
    for s in range(4):                                       # four distinct synthetic stereo batches, cycled over the 33 windows
        smp = synthetic.stereo_batch(B, seed=60 + s)
        base_batches.append({kk: (v.to(cuda) if isinstance(v, torch.Tensor) and (kk.startswith('img') or kk == 'intrinsic') else v)
                             for kk, v in smp.items()})
    seq = []
    for k in range(STEPS + 1):
        smp = dict(base_batches[k % 4])
        smp['link'] = base_batches[k % 4]['link'] + k * B
        seq.append(smp)

    # ---- the benched configuration over the whole sequence, pipelined one batch ahead like bench.py
    vo_b = _make(frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16, host_glue=True, pose_channels_last=True,
                 graph_frozen=True, graph_pose='hip')
    imu = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], np.zeros(3), np.zeros(3), tr['init'], tr['gravity'],
                    tr['rgb2imu_sync'], device='cuda', denoise_model_name=None, denoise_accel=True, denoise_gyro=False)
    loop = BilevelLoop(vo_b, imu, pp.identity_SE3(), tr['init'], loss_weight=LW, batch_size=B, device='cuda')
    before = [p.detach().clone() for p in vo_b.vonet.flowPoseNet.parameters()]
    losses_b = [loop.step(seq[k], next_sample=seq[k + 1] if k + 1 < STEPS else None) for k in range(STEPS)]
    torch.cuda.synchronize()
    poses_b = np.asarray(loop.pgo_poses, dtype=np.float64)
    motions_b = np.asarray(loop.vo_motions, dtype=np.float64)
    assert poses_b.shape == (STEPS * B + 1, 7) and motions_b.shape == (STEPS * B, 7) and np.isfinite(poses_b).all()
    grads = [p.grad for p in vo_b.vonet.flowPoseNet.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    d = loop.end_epoch('vo', trainroot=str(tmp_path), epoch=1)          # optimizer.step + zero_grad + snapshot (train.py:172-198)
    moved = sum(float((p.detach() - q).abs().sum()) for p, q in zip(vo_b.vonet.flowPoseNet.parameters(), before))
    assert moved > 0.0                                                    # Adam (lr 3e-6) stepped the pose head once
    files = sorted(os.listdir(os.path.join(str(tmp_path), '1')))
    assert files == sorted(n + '.txt' for n in ('vo_pose', 'vo_motion', 'pgo_pose', 'pgo_motion', 'pgo_vel', 'imu_pose', 'imu_motion'))
    assert np.loadtxt(os.path.join(str(tmp_path), '1', 'pgo_pose.txt')).shape == (STEPS * B + 1, 7)
    del loop, vo_b

    # ---- fp32 eager networks -> oracle glue -> oracle IMU -> oracle PVGO, window by window
    vo_f = _make()
    vo_f.vonet.train()
    init = {k: np.asarray(v, dtype=np.float64) for k, v in tr['init'].items()}
    ref_poses, ref_losses = [np.concatenate([init['pos'], init['rot']])], []
    for k in range(STEPS):
        smp = seq[k]
        with torch.no_grad():
            flow, disp, pose = vo_f.vonet(smp['img0'], smp['img1'], smp['img0_norm'], smp['img0_r_norm'], smp['intrinsic'])
        edge = edge_mask(smp['img0']).cpu().numpy()             # (bit-exact against oracle/canny.py: tests/test_edge_gpu.py)
        base = torch.linalg.norm(smp['extrinsic'][:, :3], dim=1).numpy()
        o = otv.forward_glue(flow.float().cpu().numpy(), disp.float().cpu().numpy(), pose.float().cpu().numpy(), None,
                             smp['intrinsic_calib'].numpy(), base, smp['datatype'], use_kitti_coord=True, edge=edge)
        m = o['motion']
        st, end = k * B, (k + 1) * B
        args = (tr['accels'], tr['gyros'], tr['imu_dts'], tr['rgb2imu_sync'], st, end, init, tr['gravity'])
        ipos, irot, ivel = oimu.integrate(*args, False)
        dpos, drot, dvel = oimu.integrate(*args, True)
        links = np.stack([np.arange(B), np.arange(1, B + 1)], 1)
        tl, rl, nodes, v, _ = opvgo.run_pvgo(np.concatenate([ipos, irot], 1), ivel, m, links, np.full(B, 0.1), drot, dpos, dvel,
                                             loss_weight=LW, mode='banded')
        ref_poses.extend(nodes[1:])
        ref_losses.append(float(1.0 * rl.sum() + 0.1 * tl.sum()))
        q = nodes[-1][3:]
        init = dict(pos=nodes[-1][:3], rot=q / np.linalg.norm(q), vel=v[-1])
    ref_poses = np.asarray(ref_poses)

    gt = tr['gt_pos'][:STEPS * B + 1]
    ate_b, ate_r = evaluate.ate(poses_b[:, :3], gt)[0], evaluate.ate(ref_poses[:, :3], gt)[0]
    path = float(np.linalg.norm(np.diff(gt, axis=0), axis=1).sum())
    dpos = np.linalg.norm(poses_b[:, :3] - ref_poses[:, :3], axis=1)
    drot = np.linalg.norm(lie.so3_log(lie.quat_mul(lie.quat_inv(ref_poses[:, 3:]), poses_b[:, 3:])), axis=1)
    dl = max(abs(a - b) / abs(b) for a, b in zip(losses_b, ref_losses))
    print('271-frame epoch, benched bf16 pipeline vs fp32 nets + oracle back-end: ATE %.4f vs %.4f m (rel diff %.3g); PGO trajectory '
          'apart by at most %.3g m = %.3g of the %.0f m path, %.3g rad; worst window loss rel diff %.3g'
          % (ate_b, ate_r, abs(ate_b - ate_r) / ate_r, dpos.max(), dpos.max() / path, path, drot.max(), dl))
    assert abs(ate_b - ate_r) <= TOL_ATE_REL * ate_r
    assert dpos.max() <= TOL_DRIFT_POS * path
    assert drot.max() <= TOL_DRIFT_ROT
    # ... and at every point of the trajectory, relative to the distance travelled so far (the difference may grow along the path --
    # every window starts from the previous window's optimum -- but no faster than linearly: measured 8.4e-4 of the distance at the end)
    travelled = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(gt, axis=0), axis=1))])
    assert (dpos / np.maximum(travelled, 10.0)).max() <= TOL_DRIFT_POS
