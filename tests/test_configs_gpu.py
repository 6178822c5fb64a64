"""BASELINE.json configs 1, 3 and 5 as parity cases (SURVEY.md section 8d 'Configs -> concrete synthetic inputs'), at
lengths the oracle finishes in seconds: the back-end of the bilevel loop -- camera->IMU frame change, pose chaining, IMU
pre-integration in world and motion mode, PVGO per window of 8 frames, state hand-over between windows, backward of the
VO loss -- driven through BilevelLoop (HIP) against the same loop restated on the oracle.

The conv nets are replaced by a stub that emits the synthetic VO motions (random-weight nets predict noise; their
parity is pinned by tests/test_nets_cpu.py against the reference's own modules).  Config 2 is the bench workload
(`bench.py` stereo_vio) and config 4 is covered at full size by tests/test_pvgo_gpu.py.
"""
import numpy as np
import pytest
import torch

from islam_amd import evaluate, synthetic
from oracle import imu as oimu
from oracle import lie
from oracle import pvgo as opvgo

pytestmark = pytest.mark.gpu

B = 8
# name -> trajectory settings, loss weights (run_kitti.sh:5 / run_euroc.sh:5 / run_tartanair.sh:5), camera->IMU pose
CONFIGS = {
    'kitti04': dict(traj=dict(frame_dt=0.1, imu_per_frame=10, gravity=9.81), lw=(1, 0.1, 10, 0.1),
                    T_IL=[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0], windows=5),
    'euroc_mh01': dict(traj=dict(frame_dt=0.05, imu_per_frame=10, gravity=9.81, vo_sigma_t=0.01), lw=(4, 0.1, 2, 0.1),
                       T_IL=[-0.0216, -0.0647, 0.0098, 0.0077, -0.0105, -0.7018, 0.7123], windows=5),
    'tartanair_ocean': dict(traj=dict(frame_dt=0.1, imu_per_frame=10, gravity=0.0), lw=(1.5, 0.125, 1.6875, 0.025),
                            T_IL=[0.0, 0.0, 0.0, 0.5, -0.5, 0.5, -0.5], windows=4),
}


class _StubVO(torch.nn.Module):
    """Stands in for TartanVO: returns the window's camera-frame motions right-perturbed by a trainable se(3) offset."""

    class _Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.flowPoseNet = torch.nn.Linear(1, 6, bias=False, dtype=torch.float64)
            torch.nn.init.zeros_(self.flowPoseNet.weight)

    def __init__(self, cam_motions, device):
        super().__init__()
        self.vonet = self._Net().to(device)
        self.cam_motions, self.device = cam_motions, device

    def forward(self, sample):
        from islam_amd import lietensor as pp
        k = int(sample['link'][0, 0])
        m = pp.SE3(torch.tensor(self.cam_motions[k:k + B]).to(self.device))
        off = self.vonet.flowPoseNet.weight.reshape(1, 6).expand(B, 6)
        return {'motion': m @ pp.se3(off).Exp()}


def _normalise(q):
    return q / np.linalg.norm(q)


def _oracle_loop(tr, cam_motions, T_IL, lw, windows):
    """train.py:200-299 restated on the oracle (numpy, float64)."""
    init = {k: np.asarray(v, dtype=np.float64) for k, v in tr['init'].items()}
    poses, vels, losses = [np.concatenate([init['pos'], init['rot']])], [init['vel']], []
    for w in range(windows):
        st, end = w * B, (w + 1) * B
        m = lie.se3_mul(T_IL[None], lie.se3_mul(cam_motions[st:end], lie.se3_inv(T_IL)[None]))
        args = (tr['accels'], tr['gyros'], tr['imu_dts'], tr['rgb2imu_sync'], st, end, init, tr['gravity'])
        ipos, irot, ivel = oimu.integrate(*args, False)
        dpos, drot, dvel = oimu.integrate(*args, True)
        links = np.stack([np.arange(B), np.arange(1, B + 1)], 1)
        tl, rl, nodes, v, _ = opvgo.run_pvgo(np.concatenate([ipos, irot], 1), ivel, m, links, tr['dts'][st:end], drot, dpos,
                                             dvel, loss_weight=lw, mode='banded')
        poses.extend(nodes[1:])
        vels.extend(v[1:])
        losses.append((tl, rl))
        init = dict(pos=nodes[-1][:3], rot=_normalise(nodes[-1][3:]), vel=v[-1])
    return np.asarray(poses), np.asarray(vels), losses


@pytest.mark.parametrize('name', sorted(CONFIGS))
def test_windowed_backend_matches_oracle(cuda, name):
    from islam_amd import lietensor as pp
    from islam_amd.bilevel import BilevelLoop
    from islam_amd.imu_integrator import IMUModule
    cfg = CONFIGS[name]
    nwin = cfg['windows']
    F = nwin * B + 1
    tr = synthetic.car_trajectory(F, seed=21, **cfg['traj'])
    T_IL = np.asarray(cfg['T_IL'], dtype=np.float64)
    T_IL[3:] = _normalise(T_IL[3:])
    # camera-frame motions whose image under train.py:214-215 is the synthetic IMU-frame VO motion
    cam = lie.se3_mul(lie.se3_inv(T_IL)[None], lie.se3_mul(tr['vo_motions'], T_IL[None]))

    vo = _StubVO(cam, cuda)
    imu = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], np.zeros(3), np.zeros(3), tr['init'], tr['gravity'],
                    tr['rgb2imu_sync'], device='cuda', denoise_model_name=None, denoise_accel=True, denoise_gyro=False,
                    dtype=torch.float64)
    loop = BilevelLoop(vo, imu, pp.SE3(torch.tensor(T_IL)), tr['init'], loss_weight=cfg['lw'], batch_size=B, device='cuda')
    hip_losses = []
    for w in range(nwin):
        sample = {'link': torch.stack([torch.arange(B), torch.arange(1, B + 1)], 1) + w * B,
                  'dt': torch.tensor(tr['dts'][w * B:(w + 1) * B])}
        hip_losses.append(loop.step(sample))

    ref_poses, ref_vels, ref_losses = _oracle_loop(tr, cam, T_IL, cfg['lw'], nwin)
    got = np.asarray(loop.pgo_poses, dtype=np.float64)
    assert got.shape == ref_poses.shape == (F, 7)
    d = lie.se3_log(lie.se3_mul(lie.se3_inv(ref_poses), got))
    ref = np.maximum(np.linalg.norm(lie.se3_log(ref_poses), axis=-1), 1e-6)
    # north_star tolerance: 1e-4 relative on the SE(3) log (the first pose is float32-rounded by train.py's book-keeping)
    assert (np.linalg.norm(d, axis=-1) / ref).max() < 1e-4
    np.testing.assert_allclose(np.asarray(loop.pgo_vels, dtype=np.float64)[1:], ref_vels[1:], rtol=1e-6, atol=1e-6)
    for w, (tl, rl) in enumerate(ref_losses):
        want = float(loop.rot_w * rl.sum() + loop.trans_w * tl.sum())
        assert hip_losses[w] == pytest.approx(want, rel=1e-4, abs=1e-13)

    gt = tr['gt_pos']
    ate_hip, _ = evaluate.ate(got[:, :3], gt)
    ate_ref, _ = evaluate.ate(ref_poses[:, :3], gt)
    assert abs(ate_hip - ate_ref) <= 0.01 * ate_ref          # north_star: ATE within 1 % of the reference path
    # the accumulated gradient of the VO loss reached the (stub) pose head
    g = vo.vonet.flowPoseNet.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0
    loop.end_epoch()
