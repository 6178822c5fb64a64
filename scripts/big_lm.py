import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from islam_amd import ops
import bench
dev = torch.device('cuda:0')
for N in (50001, 300007):
    prob, tr = bench.build_problem(dev, N)
    prm = ops.pvgo_default_params(bench.LOSS_WEIGHT, radius=1e4)
    ws = ops.pvgo_workspace(N, dev)
    for rep in range(2):
        n, v = prob['init_nodes'].clone(), prob['init_vels'].clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res, _ = ops.pvgo_run_chain(n, v, prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'], prm, workspace=ws)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('N=%d: %d trials, %d steps, loss %.6g, %.2f ms per run, %.1f us per LM iteration, finite=%s' % (
        N, res.trials, res.steps, res.loss, dt * 1e3, dt / res.trials * 1e6, bool(torch.isfinite(n).all())), flush=True)
