"""Level-0 elimination launch (islam_pvgo_eliminate_level0) at several chain lengths: how much of its time is SIMD sharing between
sweep wavefronts (N = 5001: 834 segments x 2 sweeps + helper on 1024 SIMDs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import ops
import bench
dev = torch.device('cuda:0')
for N in (313, 626, 1251, 2501, 3751, 5001, 7501, 10001):
    prob, tr = bench.build_problem(dev, N)
    lin, _ = ops.pvgo_linearize(prob['init_nodes'], prob['init_vels'], prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'])
    Hd, Ho, rhs = ops.pvgo_build_normal(lin, prob['dts'], N, [x ** 2 for x in bench.LOSS_WEIGHT])
    _, ms, levels = ops.pvgo_solve_chain_timed(Hd.clone(), Ho, rhs, 1e-4, workspace=ops.pvgo_workspace(N, dev))
    us = bench.eliminate_l0_burst(ops, Hd, Ho, rhs, N, levels, dev)
    print('N=%5d segments=%4d  level-0 launch %.2f us' % (N, levels[0][2], us))
