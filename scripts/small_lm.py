"""run_pvgo on the reference's own problem size (one batch of 8 frames: N = 9 nodes, train.py / run_kitti.sh): wall time per call, LM
trials, kernels launched."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from islam_amd import ops
dev = torch.device('cuda:0')
for N in [int(x) for x in os.environ.get("NS", "9,17,33").split(",")]:
    prob, tr = bench.build_problem(dev, N)
    prm = ops.pvgo_default_params(bench.LOSS_WEIGHT, radius=1e4)
    ws = ops.pvgo_workspace(N, dev)
    def run():
        n, v = prob['init_nodes'].clone(), prob['init_vels'].clone()
        res, _ = ops.pvgo_run_chain(n, v, prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'], prm, workspace=ws)
        return res
    for _ in range(5):
        res = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 50
    for _ in range(R):
        res = run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / R
    print('N=%d: %.1f us per run_pvgo (%d trials, %d steps, loss %.6g): %.1f us per LM trial' % (N, el * 1e6, res.trials, res.steps, res.loss, el * 1e6 / res.trials))
