"""Predicted LM-iteration time of the sharded loop at 2 / 4 / 8 ranks from what ONE GPU can measure:
  t(P) = t_fused(N / P)                 the fused single-GPU loop on a graph of N / P nodes -- a rank's own kernels: the same launches over
                                        its stretch of the chain (the replicated top of the tree is at most two more node steps)
       + t_shard                        what the sharded machinery adds per trial at world 1 (pack + decide launch, measured)
       + t_allreduce(bytes, P)          NOT measurable here: one latency-bound ncclAllReduce per trial over xGMI (<= 65 KB); the table
                                        carries a low / high assumption (12 / 25 us) next to the world-1 RCCL figure measured below.
Prints one JSON object; profiles/r05/scaling_model_r05.json is a committed run."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from islam_amd import ops, dist_pvgo
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
prm = ops.pvgo_default_params(bench.LOSS_WEIGHT, radius=1e4)


def fused_us(n, runs):
    prob, _ = bench.build_problem(dev, n)
    ws = ops.pvgo_workspace(n, dev)
    st = [(prob['init_nodes'].clone(), prob['init_vels'].clone()) for _ in range(runs + 2)]
    tr = 0
    for i, (a, b) in enumerate(st):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        r, _ = ops.pvgo_run_chain(a, b, prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'], prm, workspace=ws)
        if i >= 2:
            tr += r.trials
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / tr * 1e6, prob


def sharded_world1_us(prob, runs):
    tr = 0
    for i in range(runs + 2):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        _, _, r, _ = dist_pvgo.run_chain_sharded(None, prob['init_nodes'], prob['init_vels'], prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'],
                                                 prob['dts'], bench.LOSS_WEIGHT, radius=1e4, world=1, rank=0)
        if i >= 2:
            tr += r.trials
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / tr * 1e6


out = {'fused_us_per_lm_iter': {}, 'assumed_allreduce_us': [12.0, 25.0]}
for base, runs in ((5001, 60), (300007, 3)):
    for P in (1, 2, 4, 8):
        n = (base - 1) // P + 1
        us, prob = fused_us(n, runs)
        out['fused_us_per_lm_iter'][str(n)] = us
        if P == 1 and base == 5001:
            out['sharded_world1_us_per_lm_iter_N5001'] = sharded_world1_us(prob, runs)
        del prob
t_shard = out['sharded_world1_us_per_lm_iter_N5001'] - out['fused_us_per_lm_iter']['5001']
out['t_shard_us'] = t_shard
print(json.dumps(out), file=sys.stderr, flush=True)          # (partial result, in case the RCCL part below takes the process down)
table = {}
for base in (5001, 300007):
    t1 = out['fused_us_per_lm_iter'][str(base)]
    rows = {}
    for P in (2, 4, 8):
        n = (base - 1) // P + 1
        lo, hi = [out['fused_us_per_lm_iter'][str(n)] + t_shard + a for a in out['assumed_allreduce_us']]
        rows[str(P)] = {'nodes_per_rank': n, 'predicted_us': [lo, hi], 'speedup_vs_1gpu': [t1 / hi, t1 / lo]}
    table[str(base)] = {'one_gpu_us': t1, 'ranks': rows}
out['predicted'] = table
# RCCL at world 1 (a communicator of one: what the library call itself costs; no xGMI hop in it)
try:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    res = {}
    for nbytes in (8 * 1024, 64 * 1024):
        buf = torch.zeros(nbytes // 8, dtype=torch.float64, device=dev)
        for _ in range(20):
            dist.all_reduce(buf)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        res[str(nbytes)] = (time.perf_counter() - t0) / 200 * 1e6
    out['rccl_world1_allreduce_us'] = res
    dist.destroy_process_group()
except Exception as e:
    out['rccl_world1_allreduce_us'] = {'error': repr(e)[:200]}
print(json.dumps(out), flush=True)
if len(sys.argv) > 1:
    open(sys.argv[1], 'w').write(json.dumps(out, indent=1))
