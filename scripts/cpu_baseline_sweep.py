"""SURVEY section 8d CPU-baseline protocol: the oracle in PyPose-dense mode (dense J, block-diagonal W, dense J^T W J, dense
Cholesky -- what pp.optim.LM builds) and in banded mode (the same block-tridiagonal algorithm as the GPU path) over a range of
graph sizes, on this machine's host cores.  One warm-up + median of 3 runs per size; dense: the first optimizer.step only
(its iterations cost seconds), banded: the full LM loop (time per LM iteration = run time / trials)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pvgo as opvgo
from tests.helpers import chain_problem
from threadpoolctl import threadpool_limits
LW = (1, 0.1, 10, 0.1)
out = {'cores': os.cpu_count(), 'threads': {'banded': 1, 'dense': os.cpu_count()}, 'dense': {}, 'banded': {}}
for mode, sizes in (('banded', (9, 65, 257, 1025, 2049, 5001)), ('dense', (9, 65, 257, 513, 1025))):
    for N in sizes:
        prob, _ = chain_problem(N)
        ts = []
        for rep in range(4):
            # banded: many tiny numpy / LAPACK calls -- one thread is the fast setting; dense: BLAS-3 on all cores
            with threadpool_limits(limits=1 if mode == 'banded' else None):
                t0 = time.perf_counter()
                o = opvgo.run_pvgo(**prob, loss_weight=LW, mode=mode, max_steps=1 if mode == 'dense' else 10, return_optimizer=True)
                dt = time.perf_counter() - t0
            ts.append(dt / max(len(o[5].trace), 1))
        t = float(np.median(ts[1:]))
        out[mode][N] = {'s_per_lm_iter': t, 'lm_iters_per_s': 1.0 / t}
        print('%-6s N=%5d: %.4f s per LM iteration (%.2f iters/s)' % (mode, N, t, 1.0 / t), flush=True)
# cubic extrapolation of the dense path to N=5001 (needs > 60 GB: not run)
Ns = np.array(sorted(out['dense']), dtype=float)[-3:]
Ts = np.array([out['dense'][int(n)]['s_per_lm_iter'] for n in Ns])
c = np.polyfit(np.log(Ns), np.log(Ts), 1)
out['dense_extrapolated_5001'] = {'exponent': float(c[0]), 's_per_lm_iter': float(np.exp(np.polyval(c, np.log(5001.0))))}
print('dense N=5001 extrapolated (exponent %.2f): %.1f s per LM iteration' % (c[0], out['dense_extrapolated_5001']['s_per_lm_iter']))
json.dump(out, open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', 'cpu_baseline_sweep.json'), 'w'), indent=1)
