"""LM iteration time of the N=5001 benchmark graph with the sparse reprojection factor (K keypoints per link), and the
keypoint-reduction kernel alone against its HBM roofline (40 bytes read per keypoint)."""
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
for K in (0, 64, 128, 512):
    rp = None
    if K:
        z = torch.rand(N - 1, K, generator=g, dtype=torch.float64) * 30 + 5
        uv = torch.stack([torch.rand(N - 1, K, generator=g, dtype=torch.float64) * 160, torch.rand(N - 1, K, generator=g, dtype=torch.float64) * 112], -1)
        P = torch.stack([(uv[..., 0] - 80) * z / 80, (uv[..., 1] - 56) * z / 80, z], -1).to(dev).contiguous()
        tgt = (uv + torch.randn(N - 1, K, 2, generator=g, dtype=torch.float64) * 0.5 + torch.tensor([1.5, 0.0], dtype=torch.float64)).to(dev).contiguous()
        rp = ops.pvgo_reproj_struct(P, tgt, (80.0, 80.0, 80.0, 56.0), [0, 0, 0, 0.5, -0.5, 0.5, -0.5], (2.0 / K) ** 2, True)
    def run():
        n, v = prob['init_nodes'].clone(), prob['init_vels'].clone()
        res, _ = ops.pvgo_run_chain(n, v, prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'], prm, workspace=ws, reproj=rp)
        return res.trials
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr_ = 0
    for _ in range(10): tr_ += run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    line = 'K=%4d: %.1f us per LM iteration (%d trials per run)' % (K, dt / tr_ * 1e6, tr_ // 10)
    if K:
        nodes = prob['init_nodes']
        for _ in range(5): ops.pvgo_reproj_reduce(nodes, rp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): ops.pvgo_reproj_reduce(nodes, rp)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        byts = (N - 1) * K * 40 + (N - 1) * 256
        line += '   reduce kernel %.1f us = %.0f GB/s (%.1f %% of 8 TB/s)' % (us, byts / us / 1e3, byts / us / 1e3 / 80)
    print(line, flush=True)
