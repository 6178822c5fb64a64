"""Per-run wall times of islam_pvgo_run_chain on the benchmark graph (with / without the reprojection factor): outliers?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from islam_amd import ops
import bench
dev = torch.device('cuda:0')
N = 5001
prob, tr = bench.build_problem(dev, N)
ws = ops.pvgo_workspace(N, dev)
prm = ops.pvgo_default_params(bench.LOSS_WEIGHT, radius=1e4)
g = torch.Generator(device='cpu').manual_seed(0)
def mk(K):
    z = torch.rand(N - 1, K, generator=g, dtype=torch.float64) * 30 + 5
    uv = torch.stack([torch.rand(N - 1, K, generator=g, dtype=torch.float64) * 160, torch.rand(N - 1, K, generator=g, dtype=torch.float64) * 112], -1)
    P = torch.stack([(uv[..., 0] - 80) * z / 80, (uv[..., 1] - 56) * z / 80, z], -1).to(dev).contiguous()
    tgt = (uv + torch.randn(N - 1, K, 2, generator=g, dtype=torch.float64) * 0.5 + torch.tensor([1.5, 0.0], dtype=torch.float64)).to(dev).contiguous()
    return ops.pvgo_reproj_struct(P, tgt, (80.0, 80.0, 80.0, 56.0), [0, 0, 0, 0.5, -0.5, 0.5, -0.5], (2.0 / K) ** 2, True)
for K in (0, 64, 0, 128):
    rp = mk(K) if K else None
    ts = []
    for rep in range(60):
        n, v = prob['init_nodes'].clone(), prob['init_vels'].clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res, _ = ops.pvgo_run_chain(n, v, prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'], prm, workspace=ws, reproj=rp)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    ts = np.array(ts[5:])
    print('K=%3d: median %.0f us, min %.0f, p90 %.0f, max %.0f, >2x median: %d of %d' % (K, np.median(ts), ts.min(), np.percentile(ts, 90), ts.max(), (ts > 2 * np.median(ts)).sum(), len(ts)), flush=True)
