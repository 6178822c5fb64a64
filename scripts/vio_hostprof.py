"""Where the wall time of the pipelined bilevel step goes on the HOST (bench.py stereo_vio configuration): per-stage wall times of the
pipelined loop and a cProfile table of six steps."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from islam_amd import lietensor as pp, synthetic
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
steps, warmup = 12, 4
tr = synthetic.car_trajectory((steps + warmup) * batch + 1, seed=3)
imu = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], np.zeros(3), np.zeros(3), tr['init'], tr['gravity'], tr['rgb2imu_sync'],
                device=str(device), denoise_model_name=None, denoise_accel=True, denoise_gyro=False)
loop = BilevelLoop(vo, imu, pp.identity_SE3(), tr['init'], batch_size=batch, device=str(device))
samples = []
for k in range(2):
    smp = synthetic.stereo_batch(batch, seed=50 + k)
    samples.append({kk: (v.to(device) if isinstance(v, torch.Tensor) and (kk.startswith('img') or kk == 'intrinsic') else v) for kk, v in smp.items()})
seq = []
for k in range(steps + warmup + 2):
    smp = dict(samples[k % 2]); smp['link'] = samples[k % 2]['link'] + k * batch; seq.append(smp)
if os.environ.get('CACHE_FROZEN') == '1':          # the main chain alone: the frozen nets' replay replaced by cached outputs
    imgs = [samples[0][k] for k in ('img0', 'img1', 'img0_norm', 'img0_r_norm')]
    cached = tuple(t.clone() for t in vo.vonet.frozen_forward(*imgs))
    torch.cuda.synchronize()
    vo.vonet.frozen_forward = lambda *a: cached
    vo.vonet._frozen_graphed = lambda imgs_: cached
loop.reset()
for k in range(warmup):
    loop.step(seq[k], next_sample=seq[k + 1])
torch.cuda.synchronize()
loop.timing = dict(vo=0.0, imu=0.0, pgo=0.0, opt=0.0)
pr = cProfile.Profile()
t0 = time.perf_counter()
if os.environ.get('NOPROF') != '1': pr.enable()
for k in range(warmup, warmup + steps):
    loop.step(seq[k], next_sample=seq[k + 1])
if os.environ.get('NOPROF') != '1': pr.disable()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print('pipelined: %.2f ms per batch (with cProfile on); stage wall ms per batch: %s' % (el / steps * 1e3, {k: round(v / steps * 1e3, 2) for k, v in loop.timing.items()}))
if os.environ.get('NOPROF') != '1':
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(os.environ.get('SORT', 'cumulative')).print_stats(45)
    print(s.getvalue()[:9000])
