"""Randomized stress of the partitioned block-tridiagonal solver (islam_pvgo_solve_chain) against LAPACK's banded Cholesky:
random sizes 2 .. 2500, random segment lengths on both tree levels, random damping."""
import numpy as np
import pytest
import scipy.linalg as sla
import torch

pytestmark = pytest.mark.gpu


def _case(case, rng0, dev):
    from islam_amd import ops
    N = int(rng0.integers(2, 2500))
    seg = (0, 0) if case % 3 == 0 else (int(rng0.integers(0, 9)), int(rng0.integers(0, 9)))
    rng = np.random.default_rng(case)
    Hd, Ho = np.zeros((N, 9, 9)), np.zeros((N, 9, 9))
    Hd[:, np.arange(9), np.arange(9)] = rng.uniform(0.1, 2.0, (N, 9))
    Jk = rng.normal(size=(N - 1, 12, 18))
    JJ = np.einsum('kri,krj->kij', Jk, Jk)
    Hd[:-1] += JJ[:, :9, :9]
    Hd[1:] += JJ[:, 9:, 9:]
    Ho[:N - 1] = JJ[:, :9, 9:]
    rhs, damping = rng.normal(size=(N, 9)), float(rng.uniform(0, 1))
    t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    dx = ops.pvgo_solve_chain(t(Hd), t(Ho), t(rhs), damping, seg_len=seg).cpu().numpy()
    ab = np.zeros((18, 9 * N))
    for r in range(9):
        for c in range(9):
            if r >= c:
                ab[r - c, c::9] = Hd[:, r, c] * ((1 + damping) if r == c else 1.0)
            ab[9 + c - r, r:9 * (N - 1):9] = Ho[:N - 1, r, c]
    ref = sla.solveh_banded(ab, rhs.reshape(-1), lower=True).reshape(N, 9)
    return np.abs(dx - ref).max() / np.abs(ref).max(), N, seg


def test_random_sizes_and_segment_plans(cuda):
    rng0 = np.random.default_rng(12345)
    for case in range(150):
        err, N, seg = _case(case, rng0, cuda)
        assert err <= 1e-9, (case, N, seg, err)
