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
