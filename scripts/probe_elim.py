import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import islam_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), 'libislam_probe.so')
import numpy as np, torch
from islam_amd import ops
dev = torch.device('cuda:0')
N = 5001
rng = np.random.default_rng(0)
Hd = np.zeros((N, 9, 9)); Ho = np.zeros((N, 9, 9))
for k in range(N): Hd[k] += np.diag(rng.uniform(0.5, 2.0, 9))
J = rng.normal(size=(N - 1, 12, 18)) * 0.3
for k in range(N - 1):
    JJ = J[k].T @ J[k]; Hd[k] += JJ[:9, :9]; Hd[k + 1] += JJ[9:, 9:]; Ho[k] = JJ[:9, 9:]
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
Hd_d, Ho_d, rhs = t(Hd), t(Ho), t(rng.normal(size=(N, 9)))
for _ in range(5): ops.pvgo_solve_chain(Hd_d.clone(), Ho_d, rhs, 1e-4)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 1024)()
fn = L.lib()._cdll.islam_probe_read
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
b = list(buf)
print('prologue->loop start (cycles from slot0): n/a; per node phases in shader cycles (2.4 GHz):')
for tnode in range(5):
    s = [b[8 * tnode + i] for i in range(1, 6)]
    nxt = b[8 * (tnode + 1) + 1] if tnode < 4 else b[100]
    print('node %d: issue+elim %5d  stores+X %5d  schur %5d  combine %5d  update->next %5d   total %5d' % (
        tnode, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], nxt - s[4], nxt - s[0]))
