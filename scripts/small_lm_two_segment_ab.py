"""small_lm_kernel: two segments around a separator (default) against the one-segment sweep (ISLAM_SMALL_LM_ONE_SEGMENT=1): same traces,
iterates to rounding, time per run_pvgo on the 9-node window."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from islam_amd import ops
from tests.helpers import chain_problem
from tests.test_pvgo_gpu import _noisy_problem, _dev, LW
dev = torch.device('cuda:0')
def run(prob, env):
    os.environ['ISLAM_SMALL_LM_ONE_SEGMENT'] = env
    nodes, vels, poses, drots, dtrans, dvels, dts = _dev(prob, dev)
    prm = ops.pvgo_default_params(LW, radius=1e4)
    res, trace = ops.pvgo_run_chain(nodes, vels, poses, drots, dtrans, dvels, dts, prm, trace_cap=256)
    return res, np.asarray(trace), nodes.cpu().numpy(), vels.cpu().numpy()
for F, seed, sig in [(9, 0, 0.0), (9, 1, 1.5), (12, 8, 1.5), (5, 2, 1.0), (16, 3, 0.8), (9, 4, 3.0), (2, 1, 0.5), (3, 1, 0.5), (4, 1, 0.5), (8, 2, 1.0), (15, 2, 1.0)]:
    prob = _noisy_problem(F, seed, sig) if sig > 0 else chain_problem(F)[0]
    (r0, t0, n0, v0), (r1, t1, n1, v1) = run(prob, '1'), run(prob, '0')
    print(F, seed, sig, 'trials', r0.trials, r1.trials, 'steps', r0.steps, r1.steps, 'status', r0.status, r1.status,
          'loss %.12e %.12e' % (r0.loss, r1.loss), 'maxdiff nodes %.2e vels %.2e' % (np.abs(n0 - n1).max(), np.abs(v0 - v1).max()),
          'trace equal', t0.shape == t1.shape and np.allclose(t0, t1, rtol=1e-9, atol=0))
prob = chain_problem(9)[0]
for env in ('1', '0'):
    os.environ['ISLAM_SMALL_LM_ONE_SEGMENT'] = env
    nodes, vels, poses, drots, dtrans, dvels, dts = _dev(prob, dev)
    prm = ops.pvgo_default_params(LW, radius=1e4)
    ws = ops.pvgo_workspace(9, dev)
    n0, v0 = nodes.clone(), vels.clone()
    for _ in range(5):
        nodes.copy_(n0); vels.copy_(v0)
        res, _ = ops.pvgo_run_chain(nodes, vels, poses, drots, dtrans, dvels, dts, prm, workspace=ws)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50):
        nodes.copy_(n0); vels.copy_(v0)
        res, _ = ops.pvgo_run_chain(nodes, vels, poses, drots, dtrans, dvels, dts, prm, workspace=ws)
    torch.cuda.synchronize()
    print('ISLAM_SMALL_LM_ONE_SEGMENT=%s: %.1f us per run_pvgo (N=9, %d trials)' % (env, (time.perf_counter() - t) / 50 * 1e6, res.trials))
