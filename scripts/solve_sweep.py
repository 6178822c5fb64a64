"""Per-launch timing of the block-tridiagonal solve as a function of the level-0 segment length (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from islam_amd import ops
dev = torch.device('cuda:0')
N = 5001
rng = np.random.default_rng(0)
Hd = np.zeros((N, 9, 9)); Ho = np.zeros((N, 9, 9))
for k in range(N): Hd[k] += np.diag(rng.uniform(0.5, 2.0, 9))
J = rng.normal(size=(N - 1, 12, 18)) * 0.3
for k in range(N - 1):
    JJ = J[k].T @ J[k]; Hd[k] += JJ[:9, :9]; Hd[k + 1] += JJ[9:, 9:]; Ho[k] = JJ[:9, 9:]
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
Hd_d, Ho_d, rhs = t(Hd), t(Ho), t(rng.normal(size=(N, 9)))
ws = ops.pvgo_workspace(N, dev)
for seg in [(4, 0), (8, 0), (16, 0), (32, 0), (64, 0), (128, 0), (0, 0)]:
    acc = None
    for i in range(23):
        _, ms, lv = ops.pvgo_solve_chain_timed(Hd_d.clone(), Ho_d, rhs, 1e-4, seg_len=seg, workspace=ws)
        if i >= 3: acc = dict(ms) if acc is None else {k: acc[k] + ms[k] for k in ms}
    us = {k: round(a / 20 * 1e3, 1) for k, a in acc.items()}
    # whole-solve wall time without per-launch events
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    H2 = [Hd_d.clone() for _ in range(20)]
    e0.record()
    for i in range(20): ops.pvgo_solve_chain(H2[i], Ho_d, rhs, 1e-4, seg_len=seg, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    print(seg, lv, 'per-launch us', us, 'sum', round(sum(us.values()), 1), 'solve wall us (incl. sync+flag copy)', round(e0.elapsed_time(e1) / 20 * 1e3, 1))
