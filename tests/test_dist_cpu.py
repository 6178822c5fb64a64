"""world_size=2 (and 3) gloo test of the sharded PVGO driver on CPU: the collective pattern, the shard plan and the
replicated LM control of islam_amd/dist_pvgo.py, with the numpy/oracle backend standing in for the HIP kernels.
The result must equal the single-process oracle LM on the whole graph."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from islam_amd import dist_pvgo
from islam_amd.lm_control import LMControl
from oracle import pvgo as opvgo
from tests.helpers import chain_problem
from tests.np_shard_backend import NumpyBackend, plan_levels

LW = (1, 0.1, 10, 0.1)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, F, seg, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    prob, _ = chain_problem(F)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    solver = dist_pvgo.ShardedChainPVGO(t(prob['init_nodes']), t(prob['init_vels']), t(prob['vo_motions']), t(prob['imu_drots']),
                                        t(prob['imu_dtrans']), t(prob['imu_dvels']), t(prob['dts']), LW, radius=1e4,
                                        seg_len=seg, backend=NumpyBackend())
    r = solver.run()
    np.savez(os.path.join(out_dir, 'r%d.npz' % rank), nodes=r['nodes'].numpy(), vels=r['vels'].numpy(),
             trace=np.array([(a, b, float(c)) for a, b, c in r['trace']]), trials=r['trials'])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,F,seg', [(2, 120, (0, 0)), (3, 97, (5, 4)), (2, 60, (7, 0))])
def test_sharded_lm_gloo(tmp_path, world, F, seg):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, F, seg, str(tmp_path)), nprocs=world, join=True)
    prob, _ = chain_problem(F)
    ref = opvgo.run_pvgo(**prob, loss_weight=LW, mode='banded', return_optimizer=True)[5]
    outs = [np.load(os.path.join(str(tmp_path), 'r%d.npz' % r)) for r in range(world)]
    for o in outs:                                        # every rank holds the full, identical solution
        np.testing.assert_array_equal(o['nodes'], outs[0]['nodes'])
        np.testing.assert_array_equal(o['trace'], outs[0]['trace'])
    assert int(outs[0]['trials']) == len(ref.trace)
    np.testing.assert_array_equal(outs[0]['trace'][:, 2], [float(t[2]) for t in ref.trace])
    np.testing.assert_allclose(outs[0]['trace'][:, 0], [t[0] for t in ref.trace], rtol=1e-8)
    np.testing.assert_allclose(outs[0]['nodes'], ref.nodes, atol=1e-8)
    np.testing.assert_allclose(outs[0]['vels'], ref.vels, atol=1e-8)


def test_shard_plan_covers_chain_exactly_once():
    for N, seg, world in [(5001, (0, 0), 8), (5001, (19, 15), 4), (200, (4, 0), 7), (100, (9, 0), 2), (4000, (0, 0), 3)]:
        lv = plan_levels(N, seg)
        n, m, P = lv[0]
        sh = dist_pvgo.shard_plan(N, lv[0], world)
        assert sum(s['nseg'] for s in sh) == P and sh[0]['seg0'] == 0
        owned = np.zeros(N - 1, int)
        nodes = np.zeros(N, int)
        for s in sh:
            owned[s['node0']:s['node0'] + s['n_own_links']] += 1
            lo = s['node0'] + (1 if s['has_left'] else 0)
            nodes[lo:s['node0'] + s['n_own_links'] + 1] += 1
            assert (s['node0'] == 0) == (not s['has_left'])
            if s['has_left']:
                assert (s['node0'] + 1) % (m + 1) == 0             # rank boundaries sit on separator nodes
        assert np.all(owned == 1) and np.all(nodes == 1)


def test_lm_control_matches_oracle_trust_region():
    """Host control (sharded path) vs the oracle's TrustRegion/StopOnPlateau on a scripted loss sequence."""
    ctl = LMControl(radius=1e4)
    tr = opvgo.TrustRegion(radius=1e4)
    ctl.set_initial_loss(10.0)
    ctl.begin_step()
    seq = [(12.0, -3.0), (11.0, -2.5), (9.0, -1.5)]
    last = 10.0
    for loss, q in seq:
        kept = ctl.after_trial(loss, q)
        JD, R = np.array([1.0]), np.array([(q - 1.0) / 2.0])      # JD.(2R+JD) = q
        tr.update(last, loss, JD, R)
        assert np.isclose(ctl.damping, tr.pg['damping']) and np.isclose(ctl.down, tr.pg['down'])
        assert kept == (not (last < loss))
    ctl.end_step()
    assert ctl.loss == 9.0 and ctl.reject_count == 2 and ctl.steps == 1 and ctl.continual


def _grad_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from islam_amd.dist_train import allreduce_gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    net[2].bias.requires_grad_(False)
    x = torch.full((4, 7), float(rank + 1))
    net(x).sum().backward()
    net[0].bias.grad = None                                   # a parameter that saw no gradient on this rank
    nb = allreduce_gradients(net.parameters(), average=False, bucket_bytes=64)
    torch.save({'g': [None if p.grad is None else p.grad.clone() for p in net.parameters()], 'nb': nb},
               os.path.join(out_dir, 'g%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_gloo(tmp_path):
    """Config-5 exchange: bucketed sum of the pose-head gradients over 2 ranks equals the single-process sum."""
    port = _free_port()
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), 'g%d.pt' % r)) for r in range(2)]
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    tot = None
    for r in range(2):
        net.zero_grad()
        net(torch.full((4, 7), float(r + 1))).sum().backward()
        g = [p.grad.clone() for p in net.parameters()]
        g[1] = torch.zeros_like(g[1])
        tot = g if tot is None else [a + b for a, b in zip(tot, g)]
    assert outs[0]['nb'] > 1
    for r in range(2):
        for got, want, p in zip(outs[r]['g'], tot, range(4)):
            if p == 3:
                continue                                       # frozen parameter: untouched
            torch.testing.assert_close(got, want)


class _StubFrontEnd(torch.nn.Module):
    """A TartanVO stand-in: a BN'd 'stereo' branch and a per-frame 'pose' that depends on it."""

    class _V(torch.nn.Module):
        def __init__(self):
            super().__init__()
            from islam_amd import nets
            self.stereoNet = nets.StereoNet7()
            self.flowPoseNet = torch.nn.Linear(1, 7)

    def __init__(self):
        super().__init__()
        self.vonet = self._V()

    def forward(self, sample):
        disp = self.vonet.stereoNet(torch.cat([sample['img0_norm'], sample['img0_r_norm']], 1))[0]
        feat = disp.mean((1, 2, 3)).reshape(-1, 1)
        return {'motion': self.vonet.flowPoseNet(feat), 'disp': disp}


def _sample(B, H=256, W=256):
    g = torch.Generator().manual_seed(3)
    return {'img0': torch.rand(B, 3, H, W, generator=g), 'img0_norm': torch.randn(B, 3, H, W, generator=g),
            'img0_r_norm': torch.randn(B, 3, H, W, generator=g), 'datatype': ['kitti'] * B}


def _frontend_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from islam_amd.dist_train import FrameParallelVO, ShardedBatchNorm2d
    torch.manual_seed(0)
    vo = _StubFrontEnd()
    keys = list(vo.vonet.state_dict().keys())
    fp = FrameParallelVO(vo)
    assert list(vo.vonet.state_dict().keys()) == keys                   # checkpoint layout unchanged
    assert sum(isinstance(m, ShardedBatchNorm2d) for m in vo.modules()) == 34          # 34 BN layers
    vo.train()
    with torch.no_grad():
        res = fp(_sample(4))
    rm = vo.vonet.stereoNet.state_dict()
    bn = {k: v.clone() for k, v in rm.items() if 'running_' in k}
    torch.save({'motion': res['motion'].tensor().clone(), 'disp': res['disp'].clone(), 'bn': bn}, os.path.join(out_dir, 'f%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_parallel_frontend_gloo(tmp_path):
    """Section 8e row 3: frames sharded over 2 ranks with synchronised BatchNorm statistics == the un-sharded forward
    (outputs, all-gathered motions, and the running statistics that end up in the checkpoint)."""
    port = _free_port()
    mp.spawn(_frontend_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), 'f%d.pt' % r)) for r in range(2)]
    torch.manual_seed(0)
    vo = _StubFrontEnd()
    vo.train()
    with torch.no_grad():
        ref = vo(_sample(4))
    for r in range(2):
        torch.testing.assert_close(outs[r]['motion'], ref['motion'], rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(outs[r]['disp'], ref['disp'][r::2], rtol=2e-3, atol=2e-4)
        for k, v in outs[r]['bn'].items():
            torch.testing.assert_close(v, vo.vonet.stereoNet.state_dict()[k], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(outs[0]['motion'], outs[1]['motion'], rtol=0, atol=0)
