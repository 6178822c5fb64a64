"""Twisted vs one-sided elimination on the N=5001 benchmark graph: per-launch times of one solve and LM iteration time
for pinned segment lengths (ISLAM_PVGO_ONESIDED=1 selects the one-sided path for the whole process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from islam_amd import ops
import bench
dev = torch.device('cuda:0')
N = int(os.environ.get('N', 5001))
prob, tr = bench.build_problem(dev, N)
ws = ops.pvgo_workspace(N, dev)
g = torch.Generator().manual_seed(0)
Hd = torch.eye(9, dtype=torch.float64).repeat(N, 1, 1) * 20 + 0.1 * torch.randn(N, 9, 9, generator=g, dtype=torch.float64)
Hd = (Hd + Hd.transpose(1, 2)).contiguous().to(dev)
Ho = (0.3 * torch.randn(N, 9, 9, generator=g, dtype=torch.float64)).to(dev)
rhs = torch.randn(N, 9, generator=g, dtype=torch.float64).to(dev)
SWEEP = ((0, 0), (5, 5), (7, 7), (7, 5), (5, 7), (4, 4), (6, 6)) if os.environ.get('FULL') else ((0, 0), (0, 0))
for sl in SWEEP:
    best = None
    for _ in range(30):
        dx, ms, levels = ops.pvgo_solve_chain_timed(Hd.clone(), Ho, rhs, 1e-4, seg_len=sl, workspace=ws)
        tot = sum(ms.values())
        if best is None or tot < best[0]:
            best = (tot, ms)
    prm = ops.pvgo_default_params(bench.LOSS_WEIGHT, radius=1e4, seg_len=sl)
    def run():
        n, v = prob['init_nodes'].clone(), prob['init_vels'].clone()
        res, _ = ops.pvgo_run_chain(n, v, prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'], prm, workspace=ws)
        return res.trials
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr_ = 0
    for _ in range(30): tr_ += run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('seg_len=%s levels=%s: %.1f us/LM iter; solve launches (us): %s' % (
        sl, [(n, m) for n, m, P in levels], dt / tr_ * 1e6, {k: round(v * 1e3, 1) for k, v in best[1].items()}), flush=True)
