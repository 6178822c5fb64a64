"""Which chain bounds the pipelined bilevel step?  The step as benched, then the same loop with the frozen nets' replay replaced by
cached outputs (what the main chain -- pose head, glue, IMU, PVGO, backward -- costs on its own), then the frozen replays alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from islam_amd import lietensor as pp, synthetic, nets
from islam_amd.TartanVO import TartanVO
from islam_amd.bilevel import BilevelLoop
from islam_amd.imu_integrator import IMUModule
from islam_amd.miopen_pin import use_pinned_db
use_pinned_db()
device = torch.device('cuda:0')
batch = 8
torch.manual_seed(0)
vo = TartanVO(correct_scale=False, fix_parts=("flow", "stereo"), use_kitti_coord=True, frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16,
              host_glue=True, miopen_find=True, pose_channels_last=True, graph_frozen=True, graph_pose='accumulate')
with torch.no_grad():
    vo.vonet.stereoNet.conv_c13.weight.zero_(); vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
steps, warmup = 48, 6
tr = synthetic.car_trajectory((7 * (steps + warmup) + 8) * batch + 1, seed=3)
imu = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], np.zeros(3), np.zeros(3), tr['init'], tr['gravity'], tr['rgb2imu_sync'],
                device=str(device), denoise_model_name=None, denoise_accel=True, denoise_gyro=False)
loop = BilevelLoop(vo, imu, pp.identity_SE3(), tr['init'], batch_size=batch, device=str(device))
samples = []
for k in range(2):
    smp = synthetic.stereo_batch(batch, seed=50 + k)
    samples.append({kk: (v.to(device) if isinstance(v, torch.Tensor) and (kk.startswith('img') or kk == 'intrinsic') else v) for kk, v in smp.items()})
def make_seq(k0, n):
    out = []
    for k in range(k0, k0 + n):
        smp = dict(samples[k % 2]); smp['link'] = samples[k % 2]['link'] + k * batch; out.append(smp)
    return out

def run(tag, k0, pipelined=True):
    seq = make_seq(k0, steps + warmup + 2)
    nxt = (lambda k: seq[k + 1]) if pipelined else (lambda k: None)
    for k in range(warmup):
        loop.step(seq[k], next_sample=nxt(k))
    torch.cuda.synchronize()
    for k_ in loop.timing: loop.timing[k_] = 0.0
    t0 = time.perf_counter()
    for k in range(warmup, warmup + steps):
        loop.step(seq[k], next_sample=nxt(k))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    print('%-46s %.2f ms per batch (%.0f frames/s); host per stage: %s' % (tag, wall, batch / wall * 1e3, ', '.join('%s %.2f' % (k_, v_ / steps * 1e3) for k_, v_ in loop.timing.items())), flush=True)

loop.reset()
run('sequential schedule (first, like bench.py):', 0, pipelined=False)
run('pipelined step as benched:', steps + warmup)
# the frozen replay alone, back to back (no main chain)
imgs = [samples[0][k] for k in ('img0', 'img1', 'img0_norm', 'img0_r_norm')]
for _ in range(3): vo.vonet.frozen_forward(*imgs)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): vo.vonet.frozen_forward(*imgs)
torch.cuda.synchronize()
print('%-46s %.2f ms per batch' % ('frozen replay alone, back to back:', (time.perf_counter() - t0) / steps * 1e3), flush=True)
# sensitivities: the pipelined step with parts of the work removed
cached = tuple(t.clone() for t in vo.vonet.frozen_forward(*imgs))
torch.cuda.synchronize()
k0 = 2 * (steps + warmup)
orig_run_frozen = vo.vonet._run_frozen
def without(name):
    def rf(nm, master, dtype, x, quarter=False):
        if nm == name:
            return ((cached[0],), None) if nm == 'flow' else (cached[1], None)
        return orig_run_frozen(nm, master, dtype, x, quarter=quarter)
    return rf
for name in ('flow', 'stereo'):
    vo.vonet._run_frozen = without(name)
    vo.vonet.reset_graphs()
    run('pipelined, %s net replaced by a cached result:' % name, k0); k0 += steps + warmup
vo.vonet._run_frozen = orig_run_frozen
vo.vonet.reset_graphs()
# the main chain alone: cached frozen outputs
vo.vonet.frozen_forward = lambda *a: cached
vo.vonet._frozen_graphed = lambda imgs_: cached
run('main chain alone (frozen outputs cached):', k0); k0 += steps + warmup
