"""World-1 timing of the sharded C loop (islam_pvgo_run_chain_sharded, no communicator) against the single-GPU loop
(islam_pvgo_run_chain) on bench.py's 5000-frame graph: alternating runs, wall time per LM trial.  Under rocprofv3 --kernel-trace the
last run of each kind gives the launch-by-launch timeline (scripts/sharded_world1_timeline.py).
    python scripts/sharded_world1_time.py [runs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from islam_amd import dist_pvgo, ops

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda:0')
prob, _ = bench.build_problem(dev, 5000)
args = [prob[k] for k in ('init_nodes', 'init_vels', 'vo', 'drots', 'dtrans', 'dvels', 'dts')]
LW = bench.LOSS_WEIGHT
prm = ops.pvgo_default_params(LW, radius=1e4)


def single():
    n, v = args[0].clone(), args[1].clone()
    res, _ = ops.pvgo_run_chain(n, v, *args[2:], prm)
    return res.trials


def sharded():
    return dist_pvgo.run_chain_sharded(None, *args, LW)[2].trials


for f in (single, sharded):
    for _ in range(5):
        f()
out = {}
for rep in range(3):
    for name, f in (('single', single), ('sharded world 1', sharded)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr = 0
        for _ in range(runs):
            tr += f()
        torch.cuda.synchronize()
        out.setdefault(name, []).append((time.perf_counter() - t0) / tr * 1e6)
for k, v in out.items():
    print('%-16s us per LM trial: %s' % (k, ' '.join('%.1f' % x for x in v)))
print('ratio (best of each): %.3f' % (min(out['sharded world 1']) / min(out['single'])))
