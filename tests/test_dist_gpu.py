"""GPU test of the sharded PVGO path: 1/2/4/8 'virtual ranks' driven in lock-step on one MI355X (the all-reduce is a
plain sum, islam_amd.dist_pvgo.run_lockstep) must reproduce the single-GPU LM loop of islam_pvgo_run_chain."""
import numpy as np
import pytest
import torch

from tests.helpers import chain_problem

pytestmark = pytest.mark.gpu
LW = (1, 0.1, 10, 0.1)


@pytest.mark.parametrize('world,F,seg', [(1, 300, (0, 0)), (2, 300, (0, 0)), (4, 1000, (0, 0)), (8, 5001, (0, 0)), (3, 257, (5, 4)), (5, 5001, (0, 0)),
                                         (2, 5001, (0, 0)), (8, 1000, (0, 0)), (3, 65, (0, 0)), (4, 33, (0, 0))])      # the last two: exchange level 0
def test_sharded_equals_single_gpu(cuda, world, F, seg):
    from islam_amd import dist_pvgo, ops
    prob, _ = chain_problem(F)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=cuda)
    args = [t(prob[k]) for k in ('init_nodes', 'init_vels', 'vo_motions', 'imu_drots', 'imu_dtrans', 'imu_dvels', 'dts')]
    nodes, vels = args[0].clone(), args[1].clone()
    res, trace = ops.pvgo_run_chain(nodes, vels, *args[2:], ops.pvgo_default_params(LW, radius=1e4, seg_len=seg), trace_cap=256)
    solvers = [dist_pvgo.ShardedChainPVGO(*args, LW, radius=1e4, seg_len=seg, rank=r, world=world) for r in range(world)]
    outs = dist_pvgo.run_lockstep(solvers)
    for o in outs:
        assert o['trials'] == res.trials and o['steps'] == res.steps
        np.testing.assert_array_equal([float(x[2]) for x in o['trace']], trace[:, 2])
        np.testing.assert_allclose([x[0] for x in o['trace']], trace[:, 0], rtol=1e-9)
        torch.testing.assert_close(o['nodes'], nodes, rtol=0, atol=1e-9)
        torch.testing.assert_close(o['vels'], vels, rtol=0, atol=1e-9)
    # exchange volume: the level-xl interface blocks only, not the level-0 products
    be = solvers[0].be
    per_trial = solvers[0].exchanged_doubles[-2:]
    assert per_trial == [351 * be.exchange_segments, 3 + 10 * world]
    assert be.exchange_segments >= world
    if F == 5001 and world == 8:
        assert (be.exchange_level, be.exchange_segments) == (2, 23) and per_trial[0] * 8 == 64584          # 64.6 KB (round 1: 2.34 MB)
        assert per_trial[0] < 351 * solvers[0].P0 // 30
