import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from islam_amd import ops
import scipy.linalg as sla
dev = torch.device('cuda:0')
rng0 = np.random.default_rng(12345)
worst = 0.0
for case in range(150):
    N = int(rng0.integers(2, 2500))
    seg = (0, 0) if case % 3 == 0 else (int(rng0.integers(0, 9)), int(rng0.integers(0, 9)))
    rng = np.random.default_rng(case)
    Hd = np.zeros((N, 9, 9)); Ho = np.zeros((N, 9, 9))
    for k in range(N): Hd[k] += np.diag(rng.uniform(0.1, 2.0, 9))
    Jk = rng.normal(size=(max(N - 1, 0), 12, 18))
    JJ = np.einsum('kri,krj->kij', Jk, Jk)
    Hd[:-1] += JJ[:, :9, :9]; Hd[1:] += JJ[:, 9:, 9:]; Ho[:N - 1] = JJ[:, :9, 9:]
    rhs = rng.normal(size=(N, 9)); damping = float(rng.uniform(0, 1))
    t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    dx = ops.pvgo_solve_chain(t(Hd), t(Ho), t(rhs), damping, seg_len=seg).cpu().numpy()
    ab = np.zeros((18, 9 * N))
    for r in range(9):
        for c in range(9):
            if r >= c: ab[r - c, c::9] = Hd[:, r, c] * ((1 + damping) if r == c else 1.0)
            if N > 1: ab[9 + c - r, r:9 * (N - 1):9] = Ho[:N - 1, r, c]
    ref = sla.solveh_banded(ab, rhs.reshape(-1), lower=True).reshape(N, 9)
    err = np.abs(dx - ref).max() / np.abs(ref).max()
    worst = max(worst, err)
    if err > 1e-9: print('FAIL case', case, N, seg, err)
print('150 random cases, worst relative error %.2e' % worst)
