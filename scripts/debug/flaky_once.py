"""One fresh process: benched path first, then eager (the order of tests/test_benched_frontend_gpu.py); prints the rows that differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
import test_benched_frontend_gpu as T
from islam_amd import synthetic
cuda = torch.device('cuda:0')
steps = 2
tr = synthetic.car_trajectory(steps * T.B + 1, seed=3)
seq = T._samples(cuda, steps + 2)
kw = dict(frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16, host_glue=True, pose_channels_last=True,
          graph_frozen=os.environ.get('GF', '1') == '1', graph_pose=os.environ.get('GP', '1') == '1')
depth = int(os.environ.get('DEPTH', '2'))
import islam_amd.TartanVO as TV
_orig_ss = TV.stereo_scale
scales = []
def _ss(disp, flow, pose_enu, intr4, baseline, edge, th, **kw):
    out = _orig_ss(disp, flow, pose_enu, intr4, baseline, edge, th, **kw)
    scales.append((out[0].detach().float().cpu().clone(), float(edge.float().sum()), float(flow.float().abs().sum()), float(disp.float().sum()),
                   pose_enu.tensor().detach().float().cpu().clone(), float(out[2].float().sum())))
    return out
TV.stereo_scale = _ss
def record(vo, store):
    orig = vo.vonet.forward
    def fwd(img0, img1, img0_norm, img0_r_norm, intrinsic, frozen=None):
        out = orig(img0, img1, img0_norm, img0_r_norm, intrinsic, frozen=frozen)
        store.append(tuple(t.detach().float().cpu().clone() for t in out))
        return out
    vo.vonet.forward = fwd
vo = T._make(cuda, **kw)
rec_b, rec_e = [], []
record(vo, rec_b)
loop = T._loop(vo, tr)
flows = []
for k in range(steps):
    nxt = (seq[k + 1], seq[k + 2]) if depth == 2 else (seq[k + 1] if depth == 1 else None)
    loop.step(seq[k], next_sample=nxt)
torch.cuda.synchronize()
mb = np.asarray(loop.vo_motions, dtype=np.float64)
vo_e = T._make(cuda, frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16, host_glue=True)
record(vo_e, rec_e)
loop_e = T._loop(vo_e, tr)
for k in range(steps):
    loop_e.step(seq[k])
torch.cuda.synchronize()
me = np.asarray(loop_e.vo_motions, dtype=np.float64)
d = np.abs(mb - me)
bad = sorted(set(int(r) for r, _ in np.argwhere(d > 2e-3 * np.abs(me) + 2e-5)))
print('max |diff| %.2e  rows off %s' % (d.max(), bad))

for k, (b_, e_) in enumerate(zip(rec_b, rec_e)):
    print('  step %d: flow max|diff| %.2e  disp %.2e  pose %.2e' % ((k,) + tuple(float((x - y).abs().max()) for x, y in zip(b_, e_))))

n = len(scales) // 2
for k in range(n):
    a, b = scales[k], scales[n + k]
    print('  step %d: scale max|diff| %.2e (rel %.2e)  edge-sum %s/%s  |flow|-sum rel diff %.2e  disp-sum rel diff %.2e  pose diff %.2e  mask-sum %s/%s' % (
        k, float((a[0] - b[0]).abs().max()), float(((a[0] - b[0]).abs() / b[0].abs()).max()), a[1], b[1], abs(a[2] - b[2]) / b[2], abs(a[3] - b[3]) / max(b[3], 1e-9),
        float((a[4] - b[4]).abs().max()), a[5], b[5]))
