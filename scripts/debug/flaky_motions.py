"""Intermittent mismatch of the VO motions between the benched (graphs + prefetch) path and the eager path
(tests/test_benched_frontend_gpu.py): which side varies from run to run?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import test_benched_frontend_gpu as T
from islam_amd import synthetic
cuda = torch.device('cuda:0')
steps = 2
tr = synthetic.car_trajectory(steps * T.B + 1, seed=3)
seq = T._samples(cuda, steps + 2)
mode = sys.argv[1] if len(sys.argv) > 1 else 'both'


def run_b(depth=2, **over):
    kw = dict(frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16, host_glue=True, pose_channels_last=True, graph_frozen=True, graph_pose=True)
    kw.update(over)
    vo = T._make(cuda, **kw)
    loop = T._loop(vo, tr)
    for k in range(steps):
        nxt = (seq[k + 1], seq[k + 2]) if depth == 2 else (seq[k + 1] if depth == 1 else None)
        loop.step(seq[k], next_sample=nxt)
    torch.cuda.synchronize()
    return np.asarray(loop.vo_motions, dtype=np.float64)


def run_e():
    vo = T._make(cuda, frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16, host_glue=True)
    loop = T._loop(vo, tr)
    for k in range(steps):
        loop.step(seq[k])
    torch.cuda.synchronize()
    return np.asarray(loop.vo_motions, dtype=np.float64)


def dev(a, b):
    d = np.abs(a - b)
    bad = np.argwhere(d > 2e-3 * np.abs(b) + 2e-5)
    return float(d.max()), sorted(set(int(r) for r, _ in bad))


ref_e = run_e()
for name, fn in (('eager again', run_e), ('benched depth 2', lambda: run_b(2)), ('benched depth 1', lambda: run_b(1)), ('benched no prefetch', lambda: run_b(0)),
                 ('benched depth 2, no frozen graph', lambda: run_b(2, graph_frozen=False)), ('benched depth 2, no pose graph', lambda: run_b(2, graph_pose=False))):
    res = [dev(fn(), ref_e) for _ in range(int(os.environ.get('REPS', '6')))]
    print('%-36s max |diff| vs first eager run: %s   rows off: %s' % (name, ['%.1e' % r[0] for r in res], [r[1] for r in res if r[1]]))
