"""PVGO of ONE chain graph sharded over the GPUs of a node (BASELINE configs[3]).

The reference is single-GPU; this is new design.  One process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI).  The chain is cut at separators of the EXCHANGE level xl of the partitioned block Cholesky -- the highest level
below the root that still has one segment per rank (N=5001, 8 ranks: level 2 with 23 segments; islam_pvgo_shard_ranges) -- and a rank
owns everything between two such separators, on every level below.  Per LM trial a rank
  1. eliminates levels 0 .. xl of its own sub-tree       (islam_pvgo_shard_upsweep, local)
  2. ALL-REDUCE #1: the level-xl products -- the interface blocks of J^T W J / J^T W r: separator blocks, Schur
     contributions, fill -- 351 doubles per level-xl segment, summed into a zero-initialised buffer (each row is written
     by exactly one rank): 64.6 KB at N=5001 / 8 ranks (round 1 summed the level-0 products: 2.34 MB)
  3. solves the few levels above xl redundantly, back-substitutes levels xl .. 0 of its own sub-tree
                                                         (islam_pvgo_shard_downsweep), applies the trial step to its own links
  4. ALL-REDUCE #2: [sum r^2, sum JD.(2R+JD)] + the new state of each rank's first node (halo for the
     neighbour's boundary link): 2 + 10*world doubles
  5. takes the accept/reject decision (LMControl, replicated: every rank sees identical all-reduced scalars).
Ranks cut the chain only at separator nodes, so no 9x9 block is ever split between ranks.

The compute backend is injected (``backend=``): HipBackend in production; tests/np_shard_backend.py (numpy on the
oracle) lets the collective pattern be exercised with gloo on CPU.
"""
import ctypes

import numpy as np
import torch

from .lm_control import LMControl


def shard_plan(N, plan0, world, bounds=None):
    """Contiguous split of the level-0 segments.  plan0 = (n, m, P); bounds[r] = first segment of rank r (bounds[world] = P),
    default: an even split.  Returns per-rank dicts."""
    n, m, P = plan0
    assert n == N
    stride = m + 1
    if world > P:
        raise ValueError('cannot shard %d segments over %d ranks' % (P, world))
    if bounds is None:
        bounds = [int(round(r * P / world)) for r in range(world + 1)]
    out = []
    for r in range(world):
        seg0, seg1 = bounds[r], bounds[r + 1]
        first = seg0 * stride                              # first interior node of the first local segment
        sR = (seg1 - 1) * stride + m                       # right separator of the last local segment
        has_right = sR < N
        node0 = first - 1 if seg0 > 0 else 0               # left separator included (its coupling block and its step)
        node1 = min(sR + 1, N - 1) if has_right else N - 1 # one node past the right separator (link sR builds Hd[sR])
        own1 = sR if has_right else N - 1                  # owned links: [node0, own1)
        out.append(dict(seg0=seg0, nseg=seg1 - seg0, node0=node0, node1=node1, n_own_links=own1 - node0,
                        has_left=seg0 > 0, has_right=has_right, sep_left=seg0 - 1, first_node=first, rank=r, world=world))
    return out


class HipBackend:
    """Local compute on the MI355X through the C ABI (stage-level entry points of include/islam_hip.h)."""

    def __init__(self, device):
        from . import ops
        self.ops, self.dev = ops, device
        self.flags = torch.zeros(4, dtype=torch.int32, device=device)

    def plan(self, N, seg_len):
        from ._lib import lib
        sl = (ctypes.c_int * 2)(int(seg_len[0]), int(seg_len[1]))
        from ._lib import MAX_LEVELS
        p = (ctypes.c_int * (3 * MAX_LEVELS + 1))()
        nl = lib().islam_pvgo_plan(N, sl, p)
        return [(p[3 * l], p[3 * l + 1], p[3 * l + 2]) for l in range(nl)]

    def to_local(self, x):
        return x.to(self.dev, torch.float64).contiguous()

    def linearize(self, nodes, vels, data):
        return self.ops.pvgo_linearize(nodes, vels, data['poses'], data['drots'], data['dtrans'], data['dvels'], data['dts'])[0]

    def initial_loss(self, lin, n_own):
        rows = lin[[0, 1, 2, 3, 4, 5, 24, 25, 26, 36, 37, 38, 39, 40, 41], :n_own]
        return (rows * rows).sum().reshape(1)

    def build(self, lin, data, nloc, w4):
        return self.ops.pvgo_build_normal(lin, data['dts'], nloc, w4)

    def level0_bounds(self, N, seg_len, world):
        """First level-0 segment of every rank (+ the total): ranks own whole sub-trees below the exchange level."""
        from ._lib import MAX_LEVELS, c_int, check, lib
        sl = (c_int * 2)(int(seg_len[0]), int(seg_len[1]))
        out = (c_int * (2 + 2 * MAX_LEVELS))()
        bounds, info = [], None
        for r in range(world):
            check(lib().islam_pvgo_shard_ranges(N, sl, world, r, out))
            bounds.append(out[2])
            info = (out[0], out[1])
        bounds.append(bounds[-1] + out[3])
        self.exchange_level, self.exchange_segments = info
        return bounds

    def upsweep(self, H, damping, N, seg_len, sh, scratch):
        """Levels 0 .. xl of the rank's sub-tree; returns the exchange buffer (351 doubles per level-xl segment, own rows)."""
        from ._lib import c_double, c_int, c_size_t, check, lib, ptr, stream_ptr
        Hd, Ho, rhs = H
        sl = (c_int * 2)(int(seg_len[0]), int(seg_len[1]))
        ws, nbytes = scratch['ws']
        ex = scratch['exchange']
        check(lib().islam_pvgo_shard_upsweep(ptr(Hd), ptr(Ho), ptr(rhs), c_double(damping), N, sl, sh['world'], sh['rank'], sh['node0'],
                                             ptr(ws), c_size_t(nbytes), ptr(ex), ptr(self.flags), stream_ptr(self.dev)))
        return ex

    def downsweep(self, ex, N, seg_len, sh, scratch):
        from ._lib import c_int, c_size_t, check, lib, ptr, stream_ptr
        sl = (c_int * 2)(int(seg_len[0]), int(seg_len[1]))
        ws, nbytes = scratch['ws']
        dx = scratch['dx']
        check(lib().islam_pvgo_shard_downsweep(ptr(ex), N, sl, sh['world'], sh['rank'], sh['node0'], ptr(ws), c_size_t(nbytes), ptr(dx),
                                               ptr(self.flags), stream_ptr(self.dev)))
        return dx

    def trial(self, nodes, vels, dx, data, lin, n_own, scratch):
        from ._lib import check, lib, ptr, stream_ptr
        nt, vt, part = scratch['nodes_t'], scratch['vels_t'], scratch['part']
        # lin is component-major with the LOCAL link count as stride; the kernel is given that stride through a view
        lin_own = lin if lin.shape[1] == n_own else lin[:, :n_own].contiguous()
        check(lib().islam_pvgo_trial(ptr(nodes), ptr(vels), ptr(dx), ptr(data['poses']), ptr(data['drots']), ptr(data['dtrans']),
                                     ptr(data['dvels']), ptr(data['dts']), ptr(lin_own), n_own, ptr(nt), ptr(vt), ptr(part),
                                     stream_ptr(self.dev)))
        nblk = (n_own + 63) // 64
        return nt, vt, part[:2 * nblk].view(nblk, 2).sum(0)

    def failed_flag(self):
        f = self.flags[:1].to(torch.float64)
        self.flags.zero_()
        return f[0]

    def make_scratch(self, N, nloc, n1, P0):
        from . import ops
        d = self.dev
        z = lambda *s: torch.zeros(s, dtype=torch.float64, device=d)
        ws = ops.pvgo_workspace(N, d)
        ws[0].zero_()       # product rows of OTHER ranks' segments are read (as zero contributions) when an outer separator's block is composed
        return dict(dx=z(nloc, 9), nodes_t=z(nloc, 7), vels_t=z(nloc, 3), part=z(2 * ((nloc + 63) // 64) + 2),
                    ws=ws, exchange=z(351 * self.exchange_segments))


class ShardedChainPVGO:
    """run_pvgo's LM loop (pvgo.py:168-180) on one chain graph split over ``world`` ranks."""

    def __init__(self, init_nodes, init_vels, poses, drots, dtrans, dvels, dts, loss_weight, radius=1e4, seg_len=(0, 0),
                 group=None, backend=None, rank=None, world=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.be = backend if backend is not None else HipBackend(init_nodes.device)
        self.N = init_nodes.shape[0]
        self.seg_len = seg_len
        self.levels = self.be.plan(self.N, seg_len)
        if len(self.levels) < 2:
            raise ValueError('N=%d is solved by a single wavefront; nothing to shard' % self.N)
        self.P0, self.n1 = self.levels[0][2], self.levels[1][0]
        bounds = self.be.level0_bounds(self.N, seg_len, self.world)       # None: an even split of the level-0 segments
        self.sh = shard_plan(self.N, self.levels[0], self.world, bounds)[self.rank]
        self.w4 = [float(x) ** 2 for x in loss_weight[:4]]
        self.radius = radius
        a, b = self.sh['node0'], self.sh['node1']
        L = self.be.to_local
        self.init_nodes, self.init_vels = L(init_nodes), L(init_vels)
        self.data = dict(poses=L(poses[a:b]), drots=L(drots[a:b]), dtrans=L(dtrans[a:b]), dvels=L(dvels[a:b]), dts=L(dts[a:b]))
        self.nloc = b - a + 1
        self.scratch = self.be.make_scratch(self.N, self.nloc, self.n1, self.P0)

    # ---- the algorithm as a generator: every ``yield t`` is "all-reduce (sum) t in place across the ranks"
    def _steps(self, max_steps=10, patience=3, decreasing=1e-3):
        sh, be, N = self.sh, self.be, self.N
        a, b = sh['node0'], sh['node1']
        nodes, vels = self.init_nodes[a:b + 1].clone(), self.init_vels[a:b + 1].clone()
        ctl = LMControl(radius=self.radius, max_steps=max_steps, patience=patience, decreasing=decreasing)
        n_own = sh['n_own_links']
        trials = 0
        self.exchanged_doubles = []                                     # per collective of the last run (tests: exchange volume)
        while ctl.continual:
            lin = be.linearize(nodes, vels, self.data)
            if not ctl.has_loss:
                l0 = be.initial_loss(lin, n_own)
                yield l0
                ctl.set_initial_loss(float(l0[0]))
            H = be.build(lin, self.data, self.nloc, self.w4)
            ctl.begin_step()
            while True:
                ex = be.upsweep(H, ctl.damping, N, self.seg_len, sh, self.scratch)
                self.exchanged_doubles.append(ex.numel())
                yield ex                                                # all-reduce #1: J^T W J / J^T W r interface blocks
                dx = be.downsweep(ex, N, self.seg_len, sh, self.scratch)
                nt, vt, sums = be.trial(nodes, vels, dx, self.data, lin, n_own, self.scratch)
                msg = torch.zeros(3 + 10 * self.world, dtype=torch.float64, device=sums.device)
                msg[:2] = sums
                msg[2] = be.failed_flag()                               # >0 on any rank: a pivot was not positive
                first_local = sh['first_node'] - a                      # this rank's first interior node, local index
                o = 3 + 10 * self.rank
                msg[o:o + 7] = nt[first_local]
                msg[o + 7:o + 10] = vt[first_local]
                self.exchanged_doubles.append(msg.numel())
                yield msg                                               # all-reduce #2: loss, trust-region sums, halo
                trials += 1
                host = msg[:3].tolist()                                 # the one device->host read of the trial
                if host[2] > 0:
                    ctl.solver_failed()
                    break
                accepted = ctl.after_trial(host[0], host[1])
                if accepted:
                    nodes[:n_own + 1] = nt[:n_own + 1]
                    vels[:n_own + 1] = vt[:n_own + 1]
                    if sh['has_right'] and self.rank + 1 < self.world and b > sh['node0'] + n_own:
                        h = msg[3 + 10 * (self.rank + 1):3 + 10 * (self.rank + 2)]      # neighbour's first node = my last row
                        nodes[b - a] = h[:7]
                        vels[b - a] = h[7:]
                    break
            ctl.end_step()
        # assemble the full solution on every rank (sum of zero-padded owned rows)
        full = torch.zeros(N, 10, dtype=torch.float64, device=nodes.device)
        own_rows = slice(1 if sh['has_left'] else 0, n_own + 1)                                # left separator belongs to the previous rank
        full[a + own_rows.start:a + own_rows.stop, :7] = nodes[own_rows]
        full[a + own_rows.start:a + own_rows.stop, 7:] = vels[own_rows]
        yield full
        self.result = dict(nodes=full[:, :7].contiguous(), vels=full[:, 7:].contiguous(), steps=ctl.steps, trials=trials,
                           loss=ctl.loss, trace=ctl.trace)

    def run(self, **kw):
        for t in self._steps(**kw):
            if self.world > 1:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return self.result


def run_lockstep(solvers, **kw):
    """Drive several ShardedChainPVGO instances ("virtual ranks") in one process: the all-reduce is a plain sum.
    Used by the single-GPU test of the sharded path."""
    gens = [s._steps(**kw) for s in solvers]
    while True:
        bufs = []
        for g in gens:
            try:
                bufs.append(next(g))
            except StopIteration:
                bufs.append(None)
        if all(b is None for b in bufs):
            break
        assert all(b is not None for b in bufs), 'virtual ranks diverged'
        tot = bufs[0].clone()
        for b in bufs[1:]:
            tot += b.to(tot.device)
        for b in bufs:
            b.copy_(tot.to(b.device))
    return [s.result for s in solvers]


# ---------------------------------------------------------------------------------------------- the loop in the library, on RCCL
class RcclComm:
    """An RCCL communicator of the library's own (islam_dist_comm_init): rank 0 makes the 128-byte id, torch.distributed
    broadcasts it (any backend; one-off), every rank joins.  world == 1: no communicator at all."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        from ._lib import check, lib
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.handle = ctypes.c_void_p(0)
        if self.world > 1:
            ident = torch.zeros(128, dtype=torch.uint8)
            if self.rank == 0:
                buf = (ctypes.c_ubyte * 128)()
                check(lib().islam_dist_unique_id(buf))
                ident = torch.tensor(list(buf), dtype=torch.uint8)
            if dist.get_backend(group) == 'nccl':
                ident = ident.to(device if device is not None else torch.device('cuda', torch.cuda.current_device()))
            dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            raw = (ctypes.c_ubyte * 128)(*ident.cpu().tolist())
            # ncclCommInitRank binds the communicator to the HIP current device: join on THIS rank's device, whatever the
            # caller's current device is (all ranks on device 0 would be a duplicate-GPU error or a hang)
            idx = torch.device(device).index if device is not None else None
            with torch.cuda.device(idx if idx is not None else torch.cuda.current_device()):
                check(lib().islam_dist_comm_init(raw, self.world, self.rank, ctypes.byref(self.handle)))

    def info(self):
        """(ranks, rank, device) as RCCL reports them for this communicator (islam_dist_comm_info)."""
        from ._lib import check, lib
        n, r, d = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        check(lib().islam_dist_comm_info(self.handle if self.handle else None, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)))
        return n.value, r.value, d.value

    def close(self):
        from ._lib import check, lib
        if self.handle:
            check(lib().islam_dist_comm_destroy(self.handle))
            self.handle = ctypes.c_void_p(0)


def run_chain_sharded(comm, init_nodes, init_vels, poses, drots, dtrans, dvels, dts, loss_weight, radius=1e4, seg_len=(0, 0), rank=None,
                      world=None, allreduce_cb=None, params=None, reproj=None):
    """islam_pvgo_run_chain_sharded: the sharded LM loop inside the library.  ``comm``: an RcclComm (or None with world 1);
    ``allreduce_cb``: a ctypes callback instead of RCCL (tests); ``reproj``: the sparse
    reprojection factor (ops.pvgo_reproj_struct), as in ops.pvgo_run_chain.  Returns (nodes, vels, result, bytes handed to the collectives)."""
    from . import ops
    from ._lib import PvgoResult, c_size_t, check, lib, ptr, stream_ptr
    dev = init_nodes.device
    world = (comm.world if comm is not None else 1) if world is None else world
    rank = (comm.rank if comm is not None else 0) if rank is None else rank
    N = init_nodes.shape[0]
    t = lambda x: x.to(dev, torch.float64).contiguous()
    nodes, vels = t(init_nodes).clone(), t(init_vels).clone()
    args = [t(x) for x in (poses, drots, dtrans, dvels, dts)]
    prm = params if params is not None else ops.pvgo_default_params(loss_weight, radius=radius, seg_len=seg_len)      # params: a ready islam_pvgo_params
    ws, nbytes = ops.pvgo_workspace(N, dev)
    sbytes = lib().islam_pvgo_sharded_scratch_bytes(N, world)
    scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
    res, xb = PvgoResult(), ctypes.c_longlong(0)
    if reproj is not None:                      # ops.pvgo_reproj_struct over the WHOLE graph: one keypoint set per link
        assert reproj._keep[0].shape[0] == N - 1
    common = [ptr(nodes), ptr(vels)] + [ptr(a) for a in args] + [N, ctypes.byref(prm), ctypes.byref(reproj) if reproj is not None else None,
                                                                 ptr(ws), c_size_t(nbytes), ptr(scratch), c_size_t(sbytes),
                                                                 ctypes.byref(res), ctypes.byref(xb), stream_ptr(dev)]
    if allreduce_cb is not None:
        check(lib().islam_pvgo_run_chain_sharded_cb(allreduce_cb, None, world, rank, *common))
    else:
        check(lib().islam_pvgo_run_chain_sharded(comm.handle if comm is not None else None, world, rank, *common))
    return nodes, vels, res, int(xb.value)

