"""The level tree islam_pvgo_plan builds (islam_amd/csrc/pvgo.hip: plan_levels) -- host logic, no GPU: structural invariants over a range
of sizes, the plans of the benched sizes pinned, and the end-of-round-4 rule for graphs too long for six twisted levels of equal length
(the twisted maximum of seven nodes per segment on every level, the remainder as a one-sided root)."""
import ctypes

import pytest

MAXL = 6


@pytest.fixture(scope='module', autouse=True)
def _built():
    import __graft_entry__ as g
    g.build()                          # (incremental: a no-op when the library is up to date)


def _plan(N, seg=(0, 0)):
    from islam_amd._lib import lib
    L = lib()
    L.islam_pvgo_plan.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    s = (ctypes.c_int * 2)(*seg)
    out = (ctypes.c_int * (3 * MAXL + 1))()
    nl = L.islam_pvgo_plan(N, s, out)
    return [tuple(out[3 * l:3 * l + 3]) for l in range(nl)]


@pytest.mark.parametrize('N', [2, 9, 16, 17, 65, 129, 300, 513, 1000, 5001, 9001, 20011, 40011, 100003, 131073, 200001, 300007, 1000003])
def test_level_tree_is_consistent(N):
    lv = _plan(N)
    assert 1 <= len(lv) <= MAXL and lv[0][0] == N
    for l, (n, m, P) in enumerate(lv):
        root = l == len(lv) - 1
        if root:
            assert (m, P) == (n, 1)
        else:
            assert m >= 4 and P == (n + m) // (m + 1)           # segments of m interior nodes + their right separator
            assert lv[l + 1][0] == n // (m + 1)                  # the separators are the next level's nodes


def test_benched_sizes_keep_their_plans():
    assert _plan(5001) == [(5001, 5, 834), (833, 5, 139), (138, 5, 23), (23, 5, 4), (3, 3, 1)]
    assert _plan(9) == [(9, 5, 2), (1, 1, 1)]
    assert _plan(100003) == [(100003, 7, 12501), (12500, 7, 1563), (1562, 7, 196), (195, 7, 25), (24, 7, 3), (3, 3, 1)]


@pytest.mark.parametrize('N', [300007, 1000003])
def test_long_graphs_take_twisted_segments_and_a_one_sided_root(N):
    lv = _plan(N)
    assert len(lv) == MAXL and all(m == 7 for _, m, _ in lv[:-1])          # seven = the longest segment the twisted sweeps handle
    assert lv[-1][0] == lv[-1][1] > 7                                      # what six levels cannot absorb: eliminated one-sided
    forced = _plan(N, (8, 8))                                              # (a pinned length still wins over the planner)
    assert forced[0][1] == 8 and forced[1][1] == 8
