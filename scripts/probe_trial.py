"""Wall-clock (100 MHz) probes of trial_lin_kernel inside the LM loop (libislam_probe.so, scripts/build_probe.sh)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import islam_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), 'libislam_probe.so')
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
t0 = b[400]
us = lambda x: (x - t0) / 100.0
names = ['entry', 'loads+retract', 'residuals', 'quality', 'sum+ticket', 'barrier1', 'node stores', 'jacobians', 'emit(lin+LDS)', 'barrier2', 'build+copy']
for i, nm in enumerate(names):
    print('block 40  %-14s %.2f us' % (nm, us(b[400 + i])))
print('deciding wave: start %.2f  done %.2f us (relative to block 40 entry)' % (us(b[420]), us(b[421])))
