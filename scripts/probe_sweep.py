"""Wall-clock (100 MHz) probes of bt_downsweep_kernel inside the LM loop (libislam_probe.so, scripts/build_probe.sh)."""
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
t0 = b[300]
us = lambda x: (x - t0) / 100.0
print('root: start 0, eliminated %.2f, solved %.2f, published %.2f us' % (us(b[301]), us(b[302]), us(b[303])))
for po, name in ((0, 'seg 1'), (50, 'seg P/2'), (100, 'seg P-1')):
    for li in range(4):
        o = po + 310 + 10 * li
        print('level idx %d %-8s: start %.2f influence-done %.2f presleep-done %.2f wait1 %.2f ready-seen %.2f sv-arrived %.2f xsep-loaded %.2f  solved %.2f  published %.2f' % (
            li, name, us(b[o]), us(b[o + 8]), us(b[o + 5]), us(b[o + 6]), us(b[o + 1]), us(b[o + 7]), us(b[o + 2]), us(b[o + 3]), us(b[o + 4])))
