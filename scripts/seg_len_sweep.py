"""LM iteration time of islam_pvgo_run_chain at one size under forced level-0 / level-1 segment lengths (islam_pvgo_params.seg_len;
(0, 0) = the planner's own choice).    python scripts/seg_len_sweep.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from islam_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300007
dev = torch.device('cuda:0')
prob, _ = bench.build_problem(dev, N - 1)
args = [prob[k] for k in ('init_nodes', 'init_vels', 'vo', 'drots', 'dtrans', 'dvels', 'dts')]
for seg in [(0, 0), (4, 4), (5, 5), (6, 6), (7, 7), (8, 8), (7, 5), (6, 5), (10, 10)]:
    prm = ops.pvgo_default_params(bench.LOSS_WEIGHT, radius=1e4, seg_len=seg)
    best, res = 1e9, None
    for rep in range(4):
        n, v = args[0].clone(), args[1].clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, _ = ops.pvgo_run_chain(n, v, *args[2:], prm)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / res.trials)
    print('N = %d  seg_len = %-8s %8.1f us per LM trial (%d trials, status %d, loss %.9e)' % (n.shape[0], seg, best * 1e6, res.trials, res.status, res.loss), flush=True)
