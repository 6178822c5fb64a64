"""The N = 300 007 chain of bench.py's `large_graph` leg, alone (for rocprofv3): python scripts/large_graph_run.py [N] [runs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from islam_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300007
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda:0')
prob, _ = bench.build_problem(dev, N)
prm = ops.pvgo_default_params(bench.LOSS_WEIGHT, radius=1e4)
ws = ops.pvgo_workspace(N, dev)
st = [(prob['init_nodes'].clone(), prob['init_vels'].clone()) for _ in range(runs + 1)]
def step(i):
    r, _ = ops.pvgo_run_chain(st[i][0], st[i][1], prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'], prm, workspace=ws)
    return r.trials
step(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
tr = sum(step(1 + i) for i in range(runs))
torch.cuda.synchronize()
el = time.perf_counter() - t0
print('N=%d: %d LM iterations in %.1f ms = %.1f us per iteration (%.0f it/s)' % (N, tr, el * 1e3, el / tr * 1e6, tr / el))
