"""numpy stand-in for islam_amd.dist_pvgo.HipBackend (TESTS ONLY): the same per-rank operations computed with
the oracle's arithmetic and dense linear algebra, so the sharding / collective pattern of ShardedChainPVGO can run
under gloo on CPU.  The level-0 products use the layout of include/islam_hip.h (islam_pvgo_shard_eliminate)."""
import numpy as np
import scipy.linalg as sla
import torch

from oracle import pvgo as opvgo


MAXL, TOPW, MAXTOP = 6, 1, 4


BS_PAR_MAX = 7


def plan_levels(N, seg_len=(0, 0), with_top=False, twisted=False):
    """Mirror of plan_levels() in islam_amd/csrc/pvgo.hip (same arithmetic, same tie-breaking).  ``twisted``: the plan of the
    two-sided elimination -- what islam_pvgo_plan returns; the numpy backend below keeps the one-sided tree (any tree is a
    valid partition, it only has to be the same on every rank)."""
    import math
    t_node, t_launch = 2.3, 4.0
    best, best_cost, best_top = None, 1e300, 0
    for depth in range(1, MAXL + 1):
        m_auto = max(4, int(math.ceil(math.pow(float(N), 1.0 / depth))) - 1)
        if twisted:
            if m_auto > BS_PAR_MAX and depth < MAXL:
                continue
            if m_auto % 2 == 0 and m_auto + 1 <= BS_PAR_MAX:
                m_auto += 1
        lv, n, tw = [], N, twisted
        for l in range(MAXL):
            m = m_auto
            if l < 2 and seg_len[l] > 0:
                m = max(seg_len[l], 4)
            if l == MAXL - 1 or l >= depth - 1 or m + 1 >= n or n <= (BS_PAR_MAX if twisted else 12):
                lv.append((n, n, 1))
                break
            if m > BS_PAR_MAX:
                tw = False
            lv.append((n, m, (n + m) // (m + 1)))
            n = n // (m + 1)
        top = len(lv) - 1
        while top > 0 and lv[top - 1][2] <= TOPW and (len(lv) - (top - 1)) <= MAXTOP:
            top -= 1
        if len(lv) < 2 or top != len(lv) - 1:
            tw = False
        cost = t_launch
        for l, (nn, m, _) in enumerate(lv):
            root = l == len(lv) - 1
            ltw = tw and (not root or nn <= BS_PAR_MAX)
            steps = m // 2 + 1 if (ltw and m >= 3) else m
            cost += steps * t_node + (2 * t_launch if l < top else 0.0)
        if cost < best_cost - 1e-9:
            best, best_cost, best_top = lv, cost, top
    return (best, best_top) if with_top else best


class NumpyBackend:
    def __init__(self):
        self._failed = False

    def plan(self, N, seg_len):
        return plan_levels(N, seg_len)

    def to_local(self, x):
        return torch.as_tensor(np.asarray(x), dtype=torch.float64).contiguous()

    def linearize(self, nodes, vels, data):
        n, v = nodes.numpy(), vels.numpy()
        M = n.shape[0] - 1
        edges = np.stack([np.arange(M), np.arange(1, M + 1)], 1)
        res = opvgo.residuals(n, v, edges, data['poses'].numpy(), data['drots'].numpy(), data['dtrans'].numpy(),
                              data['dvels'].numpy(), data['dts'].numpy())
        A, B = opvgo.jac_blocks(n, edges, data['poses'].numpy(), data['drots'].numpy(), res[0], res[2])
        return dict(res=res, A=A, B=B, edges=edges)

    def initial_loss(self, lin, n_own):
        return torch.tensor([sum(float(np.sum(r[:n_own] ** 2)) for r in lin['res'])], dtype=torch.float64)

    def build(self, lin, data, nloc, w4):
        inp = (lin['edges'], None, None, None, None, data['dts'].numpy())
        bl = opvgo._BandedLin(np.zeros((nloc, 7)), inp, lin['res'], lin['A'], lin['B'], w4, False)
        N = nloc
        A = np.zeros((9 * N, 9 * N))
        for u in range(18):
            A[np.arange(u, 9 * N), np.arange(0, 9 * N - u)] = bl.ab[u, :9 * N - u]
        A = A + np.tril(A, -1).T
        d = np.clip(np.diagonal(A).copy(), 1e-4, 1e32)
        np.fill_diagonal(A, d)
        return dict(A=A, b=bl.b.copy())

    def level0_bounds(self, N, seg_len, world):
        return None                                  # an even split of the level-0 segments (exchange level 0)

    def upsweep(self, H, damping, N, seg_len, sh, scratch):
        """Exchange level 0: the level-0 products are what is summed over the ranks."""
        products = scratch['products']
        products.zero_()
        self.eliminate(H, damping, N, seg_len, sh, products, scratch)
        return products

    def downsweep(self, ex, N, seg_len, sh, scratch):
        n1 = plan_levels(N, seg_len)[1][0]
        return self.backsub(self.reduced_solve(ex, N, seg_len, n1, scratch), N, seg_len, sh, scratch)

    def eliminate(self, H, damping, N, seg_len, sh, products, scratch):
        (n, m, P) = plan_levels(N, seg_len)[0]
        stride = m + 1
        A, b, a0 = H['A'], H['b'], sh['node0']
        pr = products.numpy()
        Dsep, rsep = pr[:81 * P].reshape(P, 9, 9), pr[81 * P:90 * P].reshape(P, 9)
        cL, cR = pr[90 * P:171 * P].reshape(P, 9, 9), pr[171 * P:252 * P].reshape(P, 9, 9)
        fill = pr[252 * P:333 * P].reshape(P, 9, 9)
        cgL, cgR = pr[333 * P:342 * P].reshape(P, 9), pr[342 * P:351 * P].reshape(P, 9)
        blk = lambda i, j: A[9 * (i - a0):9 * (i - a0) + 9, 9 * (j - a0):9 * (j - a0) + 9]
        vec = lambda i: b[9 * (i - a0):9 * (i - a0) + 9]
        # cumulative damping of every diagonal this rank handles (interior nodes + right separators)
        scratch['segs'] = []
        for p in range(sh['seg0'], sh['seg0'] + sh['nseg']):
            c0 = p * stride
            cnt = min(m, n - c0)
            sR = c0 + m
            nodes_damped = list(range(c0, c0 + cnt)) + ([sR] if sR < n else [])
            for k in nodes_damped:
                i = 9 * (k - a0)
                dg = np.arange(i, i + 9)
                A[dg, dg] = A[dg, dg] * (1 + damping)
            I = slice(9 * (c0 - a0), 9 * (c0 + cnt - a0))
            AII = A[I, I]
            try:
                cho = sla.cho_factor(AII)
            except np.linalg.LinAlgError:
                self._failed = True
                return
            gI = b[I]
            F = np.zeros((9 * cnt, 9))
            U = np.zeros((9 * cnt, 9))
            if p > 0:
                F[:9] = blk(c0, c0 - 1)                      # coupling (first interior, left separator)
            if sR < n:
                U[-9:] = blk(sR - 1, sR)                     # coupling (last interior, right separator)
            XF, XU, Xg = sla.cho_solve(cho, F), sla.cho_solve(cho, U), sla.cho_solve(cho, gI)
            if p > 0:
                cL[p] = F.T @ XF
                cgL[p] = F.T @ Xg
            if sR < n:
                cR[p] = U.T @ XU
                cgR[p] = U.T @ Xg
                Dsep[p] = blk(sR, sR)
                rsep[p] = vec(sR)
                if p > 0:
                    fill[p] = -(F.T @ XU)
            scratch['segs'].append((p, c0, cnt, cho, F, U, gI))

    def reduced_solve(self, products, N, seg_len, n1, scratch):
        (n, m, P) = plan_levels(N, seg_len)[0]
        pr = products.numpy()
        Dsep, rsep = pr[:81 * P].reshape(P, 9, 9), pr[81 * P:90 * P].reshape(P, 9)
        cL, cR = pr[90 * P:171 * P].reshape(P, 9, 9), pr[171 * P:252 * P].reshape(P, 9, 9)
        fill = pr[252 * P:333 * P].reshape(P, 9, 9)
        cgL, cgR = pr[333 * P:342 * P].reshape(P, 9), pr[342 * P:351 * P].reshape(P, 9)
        A = np.zeros((9 * n1, 9 * n1))
        g = np.zeros(9 * n1)
        for k in range(n1):
            D = Dsep[k] - cR[k] - (cL[k + 1] if k + 1 < P else 0)
            A[9 * k:9 * k + 9, 9 * k:9 * k + 9] = D
            g[9 * k:9 * k + 9] = rsep[k] - cgR[k] - (cgL[k + 1] if k + 1 < P else 0)
            if k + 1 < n1:
                A[9 * k:9 * k + 9, 9 * k + 9:9 * k + 18] = fill[k + 1]
                A[9 * k + 9:9 * k + 18, 9 * k:9 * k + 9] = fill[k + 1].T
        try:
            x = sla.cho_solve(sla.cho_factor(A), g)
        except np.linalg.LinAlgError:
            self._failed = True
            x = np.zeros(9 * n1)
        return torch.from_numpy(x.reshape(n1, 9))

    def backsub(self, x1, N, seg_len, sh, scratch):
        (n, m, P) = plan_levels(N, seg_len)[0]
        stride = m + 1
        a0 = sh['node0']
        nloc = sh['node1'] - a0 + 1
        dx = np.zeros((nloc, 9))
        xs = x1.numpy()
        for (p, c0, cnt, cho, F, U, gI) in scratch['segs']:
            r = gI.copy()
            if p > 0:
                r -= F @ xs[p - 1]
            sR = c0 + m
            if sR < n:
                r -= U @ xs[p]
                dx[sR - a0] = xs[p]
            dx[c0 - a0:c0 - a0 + cnt] = sla.cho_solve(cho, r).reshape(cnt, 9)
        if sh['has_left']:
            dx[0] = xs[sh['sep_left']]
        return torch.from_numpy(dx)

    def trial(self, nodes, vels, dx, data, lin, n_own, scratch):
        d = dx.numpy()
        nn, vv = opvgo.retract(nodes.numpy()[:n_own + 1], vels.numpy()[:n_own + 1], d[:n_own + 1, :6], d[:n_own + 1, 6:])
        edges = lin['edges'][:n_own]
        sl = lambda t: t.numpy()[:n_own]
        res = opvgo.residuals(nn, vv, edges, sl(data['poses']), sl(data['drots']), sl(data['dtrans']), sl(data['dvels']), sl(data['dts']))
        loss = sum(float(np.sum(r ** 2)) for r in res)
        dp = d[1:n_own + 1, :6] - d[:n_own, :6]
        A, B = lin['A'][:n_own], lin['B'][:n_own]
        dts = sl(data['dts'])
        jd = [(A @ dp[:, :, None])[:, :, 0], d[:n_own, 6:] - d[1:n_own + 1, 6:], (B @ dp[:, 3:, None])[:, :, 0],
              dp[:, :3] - dts[:, None] * d[:n_own, 6:]]
        R = [r[:n_own] for r in lin['res']]
        q = sum(float(np.sum(j * (2 * r + j))) for j, r in zip(jd, R))
        nt, vt = nodes.clone(), vels.clone()
        nt[:n_own + 1] = torch.from_numpy(nn)
        vt[:n_own + 1] = torch.from_numpy(vv)
        return nt, vt, torch.tensor([loss, q], dtype=torch.float64)

    def failed_flag(self):
        f, self._failed = self._failed, False
        return torch.tensor(1.0 if f else 0.0, dtype=torch.float64)

    def make_scratch(self, N, nloc, n1, P0):
        return dict(products=torch.zeros(351 * P0, dtype=torch.float64))
