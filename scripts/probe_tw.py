"""Wall-clock (100 MHz) probes of bt_eliminate_tw_kernel, level 1 segment 1, inside the LM loop (libislam_probe.so)."""
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
t0 = min(b[440], b[470], b[500])
us = lambda x: (x - t0) / 100.0
for name, base, steps in (('wave A (forward)', 440, 3), ('wave B (reverse)', 470, 2)):
    print('%s: entry %.2f  first loads arrived %.2f' % (name, us(b[base]), us(b[base + 1])))
    for t in range(steps):
        o = base + 2 + 5 * t
        print('   step %d: start %.2f  pivots done %.2f  schur done %.2f  barrier passed %.2f  stores issued %.2f  next formed %.2f' % (
            t, us(b[o]), us(b[o + 1]), us(b[(540 if base == 440 else 550) + t]), us(b[o + 2]), us(b[o + 3]), us(b[o + 4])))
    print('   end %.2f us' % us(b[base + 29]))

print('helper: entry %.2f  first column set written %.2f' % (us(b[500]), us(b[501])))
for t in range(3):
    o = 502 + 4 * t
    print('   iteration %d: at barrier %.2f  passed %.2f  stage streamed out %.2f  next columns written %.2f' % (
        t, us(b[o]), us(b[o + 1]), us(b[o + 2]), us(b[o + 3])))
