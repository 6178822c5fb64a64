"""LM iteration time of the N=5001 benchmark graph for pinned segment lengths of levels 0 / 1 (planner tuning)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import ops
import bench
dev = torch.device('cuda:0')
prob, tr = bench.build_problem(dev, 5001)
ws = ops.pvgo_workspace(5001, dev)
for sl in ((0, 0), (5, 5), (7, 5), (7, 7), (5, 7), (7, 0), (5, 3), (3, 5), (7, 3)):
    prm = ops.pvgo_default_params(bench.LOSS_WEIGHT, radius=1e4, seg_len=sl)
    def run():
        n, v = prob['init_nodes'].clone(), prob['init_vels'].clone()
        res, _ = ops.pvgo_run_chain(n, v, prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'], prm, workspace=ws)
        return res.trials
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr_ = 0
    for _ in range(60): tr_ += run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('seg_len=%s: %.1f us per LM iteration' % (sl, dt / tr_ * 1e6), flush=True)
