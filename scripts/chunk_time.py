"""LM iteration time of islam_pvgo_run_chain on chains beyond the size one launch of trial_elim_kernel covers with FZ_S segments per
workgroup (with ISLAM_FZ_CHUNKS=1 it loops over chunks of FZ_S segments).  Run once with ISLAM_FZ_CHUNKS=1 and once without (launch-per-stage loop).
    python scripts/chunk_time.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from islam_amd import ops

dev = torch.device('cuda:0')
for N in [int(a) for a in sys.argv[1:]] or [9001, 20011, 40011, 100003]:
    prob, _ = bench.build_problem(dev, N - 1)
    args = [prob[k] for k in ('init_nodes', 'init_vels', 'vo', 'drots', 'dtrans', 'dvels', 'dts')]
    prm = ops.pvgo_default_params(bench.LOSS_WEIGHT, radius=1e4)
    best = 1e9
    for rep in range(6):
        n, v = args[0].clone(), args[1].clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, _ = ops.pvgo_run_chain(n, v, *args[2:], prm)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / res.trials)
    print('N = %6d  ISLAM_FZ_CHUNKS=%s: %7.1f us per LM trial (%d trials, loss %.6e)' % (n.shape[0], os.environ.get('ISLAM_FZ_CHUNKS', '0'), best * 1e6, res.trials, res.loss), flush=True)
