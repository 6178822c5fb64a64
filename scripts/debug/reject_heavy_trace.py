import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, bench
from islam_amd import ops, lietensor as pp
dev = torch.device('cuda:0')
prob, tr = bench.build_problem(dev, 5001)
prm = ops.pvgo_default_params(bench.LOSS_WEIGHT, radius=1e4)
for seed, sig in ((6, 1.5), (7, 1.5), (6, 0.5), (3, 3.0)):
    g = torch.Generator().manual_seed(seed)
    N = 5001
    dt_ = torch.randn(N, 3, generator=g, dtype=torch.float64) * sig
    dr_ = torch.randn(N, 3, generator=g, dtype=torch.float64) * (0.2 * sig)
    n0 = prob['init_nodes'].cpu()
    n0 = torch.cat([n0[:, :3] + dt_, n0[:, 3:]], 1)
    pert = pp.SE3(torch.cat([torch.zeros(N, 3, dtype=torch.float64), pp.so3(dr_).Exp().tensor()], 1))
    start = (pert @ pp.SE3(n0)).tensor().to(dev).contiguous()
    res, trace = ops.pvgo_run_chain(start.clone(), prob['init_vels'].clone(), prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'], prm, trace_cap=64)
    print('seed', seed, 'sig', sig, 'trials', res.trials, 'steps', res.steps)
    for row in np.asarray(trace)[:res.trials]:
        print('   loss %.6e damping %.3e accepted %d' % (row[0], row[1], int(row[2])))
