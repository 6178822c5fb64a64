"""bench.py's second strong-scaling point (N = 300 007) through the sharded C loop with 2 and 8 ranks AS THREADS on one GPU (the
all-reduce is the host-side callback of tests/test_dist_c_gpu.py): must reproduce the fused single-GPU loop -- what `bench.py --gpus N`
runs on real RCCL.    python scripts/large_sharded_check.py [N] [worlds...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from islam_amd import ops
from tests.test_dist_c_gpu import _run_ranks_as_threads, LW

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300007
worlds = [int(a) for a in sys.argv[2:]] or [2, 8]
dev = torch.device('cuda:0')
prob, _ = bench.build_problem(dev, N)
args = [prob[k] for k in ('init_nodes', 'init_vels', 'vo', 'drots', 'dtrans', 'dvels', 'dts')]
nodes, vels = args[0].clone(), args[1].clone()
t0 = time.perf_counter()
res, _ = ops.pvgo_run_chain(nodes, vels, *args[2:], ops.pvgo_default_params(LW, radius=1e4))
torch.cuda.synchronize()
print('single GPU: trials %d steps %d status %d loss %.9e  (%.1f ms)' % (res.trials, res.steps, res.status, res.loss, (time.perf_counter() - t0) * 1e3), flush=True)
for w in worlds:
    t0 = time.perf_counter()
    outs = _run_ranks_as_threads(args, w)
    dt = time.perf_counter() - t0
    for r, (n, v, rr, xb) in enumerate(outs):
        assert (rr.trials, rr.steps, rr.status) == (res.trials, res.steps, 0), (w, r, rr.trials, rr.steps, rr.status)
        dn, dv = (n - nodes).abs().max().item(), (v - vels).abs().max().item()
        # (a 300 007-frame trajectory spans tens of kilometres: 1e-7 m is 1e-12 of the coordinates; the ranks sum the interface blocks in another order)
        assert dn <= 1e-6 and dv <= 1e-6, (w, r, dn, dv)
    print('world %d (ranks as threads): trials %d, max |dnodes| %.1e |dvels| %.1e, exchanged %d bytes per rank  (%.1f s with the host-side all-reduce)'
          % (w, outs[0][2].trials, dn, dv, outs[0][3], dt), flush=True)
print('OK')
