"""The EXACT front-end configuration bench.py times for `stereo_vio`: B=8 at
448x640, bf16 execution copies of the frozen flow / stereo nets, `graph_frozen`, `graph_pose`, the prefetch schedule
(`BilevelLoop.step(sample, next_sample=...)`), `host_glue`, `pose_channels_last` -- two bilevel steps, compared with

  (a) the fp32 eager networks driven through the ORACLE's glue, IMU pre-integration and PVGO (oracle/tartanvo.py, oracle/imu.py,
      oracle/pvgo.py): VO motions, PGO poses and the loss, within the bf16 bounds stated below;
  (b) the same bf16 frozen nets and float64 host glue with an EAGER pose head (no graphs, NCHW, no prefetch): the accumulated
      pose-head gradients must be the same numbers (fp32 kernels of different layouts: see the bound at the assertion).

`miopen_find` (MIOpen's timing-based kernel search, a ~1.5 minute one-off) is the one bench switch left off here: it changes which
fp32 kernel MIOpen picks for the pose head, nothing else.  Reference: TartanVO.py:90-198, train.py:200-299."""
import numpy as np
import pytest
import torch

from islam_amd import synthetic
from oracle import imu as oimu
from oracle import lie
from oracle import pvgo as opvgo
from oracle import tartanvo as otv

pytestmark = pytest.mark.gpu
B = 8
LW = (1, 0.1, 10, 0.1)
# bf16 operands in the two frozen nets (golden bounds: flow 3e-2, disparity 4e-2 relative to the largest value) reach the pose
# head as input noise.  Bounds on the end-to-end quantities, measured 2026-10 on MI355X with ~3x margin:
TOL_MOTION_T, TOL_MOTION_R = 8e-2, 3e-3          # VO motions: translation relative to its norm, rotation in rad
TOL_POSE = 5e-2                                  # PGO poses: |Log(ref^-1 got)| relative to max(|Log(ref)|, 1)
TOL_LOSS = 0.25                                  # rot_w * rot_loss + trans_w * trans_loss, relative
# pose-head gradients vs eager, per tensor: |diff| relative to the largest entry, and the cosine.  The two heads run different fp32
# kernels (NHWC graph replay vs NCHW eager: motions agree to 5e-4), and the bilevel loss gradient amplifies that through the PVGO
# optimum; measured 1.4e-2 / 0.99991 on the first (deepest in the backward pass) layers.  The tight graph-vs-eager check at identical
# inputs is tests/test_frontend_gpu.py::test_pose_head_graph_replay_trains_like_eager.
GRAD_TOL, GRAD_COS = 4e-2, 0.9995


def _make(cuda, **kw):
    from islam_amd.TartanVO import TartanVO
    torch.manual_seed(0)
    vo = TartanVO(correct_scale=False, fix_parts=('flow', 'stereo'), use_kitti_coord=True, **kw)
    with torch.no_grad():      # random weights predict garbage disparity: pin the stereo head to 10 px (as bench.py does)
        vo.vonet.stereoNet.conv_c13.weight.zero_()
        vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
    return vo


def _samples(cuda, n):
    out = []
    for k in range(n):
        smp = synthetic.stereo_batch(B, seed=50 + (k % 2))
        smp = {kk: (v.to(cuda) if isinstance(v, torch.Tensor) and (kk.startswith('img') or kk == 'intrinsic') else v) for kk, v in smp.items()}
        smp['link'] = smp['link'] + k * B
        out.append(smp)
    return out


def _loop(vo, tr):
    from islam_amd import lietensor as pp
    from islam_amd.bilevel import BilevelLoop
    from islam_amd.imu_integrator import IMUModule
    imu = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], np.zeros(3), np.zeros(3), tr['init'], tr['gravity'],
                    tr['rgb2imu_sync'], device='cuda', denoise_model_name=None, denoise_accel=True, denoise_gyro=False)
    return BilevelLoop(vo, imu, pp.identity_SE3(), tr['init'], batch_size=B, device='cuda')


@pytest.mark.parametrize('instances', [1, 2])
def test_benched_configuration_two_bilevel_steps(cuda, instances):
    """instances: captured copies of the frozen forward (TartanVO(graph_instances=...)).  2 = what bench.py runs: two batches ahead on
    two copies used round-robin, their replays queued back to back (four steps, so that every copy replays twice); 1 = two batches
    ahead on ONE copy, serialised by the per-copy fence."""
    steps = 2 * instances
    tr = synthetic.car_trajectory(steps * B + 1, seed=3)
    seq = _samples(cuda, steps + 2)

    # ---- the benched configuration, pipelined exactly like bench.py's timed loop
    vo_b = _make(cuda, frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16, host_glue=True, pose_channels_last=True,
                 graph_frozen=True, graph_pose='hip', graph_instances=instances)
    loop_b = _loop(vo_b, tr)
    losses_b = [loop_b.step(seq[k], next_sample=(seq[k + 1], seq[k + 2])) for k in range(steps)]      # (a tuple: two batches ahead -- the deepest schedule BilevelLoop offers; bench.py runs one ahead)
    torch.cuda.synchronize()
    motions_b = np.asarray(loop_b.vo_motions, dtype=np.float64)
    poses_b = np.asarray(loop_b.pgo_poses, dtype=np.float64)
    grads_b = [p.grad.detach().float().cpu().clone() for p in vo_b.vonet.flowPoseNet.parameters()]
    assert len(vo_b.vonet._graphs) >= 1                      # the frozen forward really replayed from a captured graph
    assert all(len(r['inst']) == instances for r in vo_b.vonet._graphs.values())
    assert all(torch.isfinite(g).all() for g in grads_b) and any(float(g.abs().sum()) > 0 for g in grads_b)

    # ---- (b) same frozen nets, eager pose head, device glue, sequential schedule: the gradients must agree
    vo_e = _make(cuda, frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16, host_glue=True)
    loop_e = _loop(vo_e, tr)
    losses_e = [loop_e.step(seq[k]) for k in range(steps)]
    grads_e = [p.grad.detach().float().cpu() for p in vo_e.vonet.flowPoseNet.parameters()]
    np.testing.assert_allclose(np.asarray(loop_e.vo_motions, dtype=np.float64), motions_b, rtol=2e-3, atol=2e-5)       # fp32 pose head, NHWC graph replay vs NCHW eager kernels
    for k in range(steps):
        assert losses_b[k] == pytest.approx(losses_e[k], rel=2e-3)
    worst, cos_min = 0.0, 1.0
    names = [n for n, _ in vo_b.vonet.flowPoseNet.named_parameters()]
    for n, gb, ge in zip(names, grads_b, grads_e):
        assert gb.shape == ge.shape
        rel = float((gb - ge).abs().max()) / max(float(ge.abs().max()), 1e-30)
        cos = float((gb * ge).sum() / (gb.norm() * ge.norm()).clamp_min(1e-30))
        if rel > 0.5 * max(worst, 1e-4):
            print('   %-40s rel %.3g cos %.8f' % (n, rel, cos))
        worst, cos_min = max(worst, rel), min(cos_min, cos)
    print('pose-head gradients, benched configuration vs eager: max |diff| / max |g| per tensor = %.3g, min cosine %.8f' % (worst, cos_min))
    assert worst <= GRAD_TOL and cos_min >= GRAD_COS
    del vo_e, loop_e

    # ---- (a) fp32 eager networks -> oracle glue -> oracle IMU -> oracle PVGO
    vo_f = _make(cuda)
    vo_f.vonet.train()
    init = {k: np.asarray(v, dtype=np.float64) for k, v in tr['init'].items()}
    ref_motions, ref_poses, ref_losses = [], [np.concatenate([init['pos'], init['rot']])], []
    for k in range(steps):
        smp = seq[k]
        with torch.no_grad():
            flow, disp, pose = vo_f.vonet(smp['img0'], smp['img1'], smp['img0_norm'], smp['img0_r_norm'], smp['intrinsic'])
        from islam_amd.edges import edge_mask
        edge = edge_mask(smp['img0']).cpu().numpy()             # (bit-exact against oracle/canny.py: tests/test_edge_gpu.py)
        base = torch.linalg.norm(smp['extrinsic'][:, :3], dim=1).numpy()
        o = otv.forward_glue(flow.float().cpu().numpy(), disp.float().cpu().numpy(), pose.float().cpu().numpy(), None,
                             smp['intrinsic_calib'].numpy(), base, smp['datatype'], use_kitti_coord=True, edge=edge)
        m = o['motion']                                          # T_IL = identity: train.py:214-215 is a no-op
        ref_motions.extend(m)
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
    ref_motions, ref_poses = np.asarray(ref_motions), np.asarray(ref_poses)
    assert motions_b.shape == ref_motions.shape == (steps * B, 7) and poses_b.shape == ref_poses.shape
    dt = np.linalg.norm(motions_b[:, :3] - ref_motions[:, :3], axis=1) / np.linalg.norm(ref_motions[:, :3], axis=1)
    dr = np.linalg.norm(lie.so3_log(lie.quat_mul(lie.quat_inv(ref_motions[:, 3:]), motions_b[:, 3:])), axis=1)
    d = np.linalg.norm(lie.se3_log(lie.se3_mul(lie.se3_inv(ref_poses), poses_b)), axis=1)
    dref = np.maximum(np.linalg.norm(lie.se3_log(ref_poses), axis=1), 1.0)
    dl = [abs(a - b) / abs(b) for a, b in zip(losses_b, ref_losses)]
    print('benched bf16 pipeline vs fp32 nets + oracle back-end: motion trans %.3g rot %.3g rad, PGO pose %.3g, loss %.3g' %
          (dt.max(), dr.max(), (d / dref).max(), max(dl)))
    assert dt.max() <= TOL_MOTION_T and dr.max() <= TOL_MOTION_R
    assert (d / dref).max() <= TOL_POSE
    assert max(dl) <= TOL_LOSS
