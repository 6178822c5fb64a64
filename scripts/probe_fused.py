"""Wall-clock (100 MHz) probes of trial_elim_kernel inside the LM loop (libislam_probe.so): three segments x three waves."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import islam_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), 'libislam_probe%s.so' % os.environ.get('ISLAM_PROBE_SUFFIX', ''))
import torch
from islam_amd import ops
import bench
dev = torch.device('cuda:0')
prob, tr = bench.build_problem(dev, 5001)
prm = ops.pvgo_default_params(bench.LOSS_WEIGHT, radius=1e4)
ws = ops.pvgo_workspace(5001, dev)
for _ in range(3):
    n, v = prob['init_nodes'].clone(), prob['init_vels'].clone()
    res, _ = ops.pvgo_run_chain(n, v, prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'], prm, workspace=ws)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 1024)()
fn = L.lib()._cdll.islam_probe_read
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
b = list(buf)
names = ['entry', 'retracted', 'links done', 'blocks built', 'elim start', 'elim end']
for seg, so in (('wg=1', 0), ('wg=nwg/2', 100)):
    t0 = min(b[600 + so + 12 * w] for w in range(10) if b[600 + so + 12 * w] > 0)
    for w in range(10):
        base = 600 + so + 12 * w
        print('%s wave %d: ' % (seg, w) + '  '.join('%s %.2f' % (names[i], (b[base + i] - t0) / 100.0) for i in range(6) if b[base + i] >= t0))

# launch ramp / drain of the grid: entry (slots 300+b) and exit (slots b) of wave 0 of workgroup b
ent = [b[300 + i] for i in range(300) if b[300 + i] > 0 and b[i] > 0]
ext = [b[i] for i in range(300) if b[300 + i] > 0 and b[i] > 0]
if ent:
    e0 = min(ent)
    ent_s, ext_s = sorted(ent), sorted(ext)
    q = lambda v, f: (v[int(f * (len(v) - 1))] - e0) / 100.0
    print('grid of %d workgroups: entry  first 0.00  median %.2f  90%% %.2f  last %.2f us' % (len(ent), q(ent_s, 0.5), q(ent_s, 0.9), q(ent_s, 1.0)))
    print('                       exit   first %.2f  median %.2f  90%% %.2f  last %.2f us' % (q(ext_s, 0.0), q(ext_s, 0.5), q(ext_s, 0.9), q(ext_s, 1.0)))
    life = sorted(x - y for x, y in zip(ext, ent))
    print('                       life   min %.2f  median %.2f  max %.2f us' % (life[0] / 100.0, life[len(life) // 2] / 100.0, life[-1] / 100.0))
