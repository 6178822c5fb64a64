"""LM iteration time at N = 5001 under forced level-0 / upper-level segment lengths (islam_pvgo_params.seg_len; (0, 0) = the planner's
choice): is the plan the planner picks still the fastest on this build?    python scripts/debug/plan_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from islam_amd import ops
N = 5001
dev = torch.device('cuda:0')
prob, _ = bench.build_problem(dev, N)
args = [prob[k] for k in ('init_nodes', 'init_vels', 'vo', 'drots', 'dtrans', 'dvels', 'dts')]
ws = ops.pvgo_workspace(N, dev)
rows = []
for m0 in (0, 4, 5, 6, 7, 8, 10):
    for m1 in (0, 4, 5, 6, 8, 12, 16, 24):
        if (m0 == 0) != (m1 == 0) and m0 == 0:
            continue
        prm = ops.pvgo_default_params(bench.LOSS_WEIGHT, radius=1e4, seg_len=(m0, m1))
        ts, res = [], None
        try:
            for rep in range(12):
                n, v = args[0].clone(), args[1].clone()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res, _ = ops.pvgo_run_chain(n, v, *args[2:], prm, workspace=ws)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / res.trials)
        except Exception as e:
            print('seg_len (%d, %d): %s' % (m0, m1, str(e)[:80]), flush=True)
            continue
        ts.sort()
        rows.append((ts[len(ts) // 2] * 1e6, m0, m1, res.trials, res.loss))
        print('seg_len = (%2d, %2d)  %7.1f us per LM trial (median of 12 synchronous runs; %d trials, loss %.9e)' % (m0, m1, rows[-1][0], res.trials, res.loss), flush=True)
rows.sort()
print('fastest:', rows[:5])
