"""Per-phase shader-clock probes of trial_kernel / linbuild_kernel inside the LM loop.
Build: hipcc -DISLAM_PROBE (scripts/build_probe.sh) -> islam_amd/lib/libislam_probe.so; run on the GPU box."""
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
buf = (ctypes.c_longlong * 512)()
L.lib().islam_probe_read.argtypes = [ctypes.c_void_p]
assert L.lib().islam_probe_read(buf) == 0
b = list(buf)
print('clock64 ticks (100 MHz constant clock on gfx950: 1 tick = 10 ns)' )
names = {200: 'entry', 201: 'loads+retract', 202: 'link_residuals', 203: 'quality', 204: 'wave_sum+part', 205: 'ticket', 206: 'lastblk entry', 207: 'control done'}
for a in range(201, 208):
    print('trial   %-16s +%6d' % (names[a], b[a] - b[a - 1]))
print('trial block1 total', b[205] - b[200], ' last block tail', b[207] - b[206])
names = {221: 'loads+residuals', 222: 'jacobians', 223: 'products+LDS', 224: 'sum+barrier', 225: 'node build+stores'}
for a in range(221, 226):
    print('linbuild %-18s +%6d' % (names[a], b[a] - b[a - 1]))
print('linbuild total', b[225] - b[220])
