"""GPU tests of the sharded LM loop INSIDE the library (islam_pvgo_run_chain_sharded, islam_amd/csrc/pvgo_dist.hip): world 1
with and without an RCCL communicator, and 2 / 3 / 8 ranks driven as threads on one GPU through the callback variant (the
all-reduce is a host-side sum across the threads) -- all must reproduce the fused single-GPU loop of islam_pvgo_run_chain."""
import ctypes
import threading

import numpy as np
import pytest
import torch

from tests.helpers import chain_problem

pytestmark = pytest.mark.gpu
LW = (1, 0.1, 10, 0.1)


def _problem(F, cuda):
    prob, _ = chain_problem(F)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=cuda)
    return [t(prob[k]) for k in ('init_nodes', 'init_vels', 'vo_motions', 'imu_drots', 'imu_dtrans', 'imu_dvels', 'dts')]


def _single(args, seg=(0, 0)):
    from islam_amd import ops
    nodes, vels = args[0].clone(), args[1].clone()
    res, trace = ops.pvgo_run_chain(nodes, vels, *args[2:], ops.pvgo_default_params(LW, radius=1e4, seg_len=seg), trace_cap=256)
    return nodes, vels, res


@pytest.mark.parametrize('F', [300, 5001])
def test_world_one_with_and_without_a_communicator(cuda, F):
    from islam_amd import dist_pvgo
    args = _problem(F, cuda)
    nodes, vels, res = _single(args)
    n1, v1, r1, xb = dist_pvgo.run_chain_sharded(None, *args, LW)
    assert (r1.trials, r1.steps, r1.status) == (res.trials, res.steps, 0)
    torch.testing.assert_close(n1, nodes, rtol=0, atol=1e-9)
    torch.testing.assert_close(v1, vels, rtol=0, atol=1e-9)
    assert r1.loss == pytest.approx(res.loss, rel=1e-9)
    # a real RCCL communicator of one rank: ncclCommInitRank / ncclCommDestroy through the C ABI
    comm = dist_pvgo.RcclComm()
    assert comm.world == 1
    from islam_amd._lib import check, lib
    ident = (ctypes.c_ubyte * 128)()
    check(lib().islam_dist_unique_id(ident))
    h = ctypes.c_void_p(0)
    check(lib().islam_dist_comm_init(ident, 1, 0, ctypes.byref(h)))
    comm.handle, comm.world = h, 1
    n2, v2, r2, _ = dist_pvgo.run_chain_sharded(comm, *args, LW, world=1, rank=0)
    comm.close()
    assert torch.equal(n2, n1) and torch.equal(v2, v1) and r2.trials == r1.trials


def _run_ranks_as_threads(args, world, params=None, reproj=None):
    """Every rank's C loop in its own thread and stream; the injected all-reduce copies each rank's buffer to the host, sums in
    rank order and writes the sum back."""
    from islam_amd import dist_pvgo
    from islam_amd._lib import lib
    hip = ctypes.CDLL('libamdhip64.so')
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    barrier = threading.Barrier(world)
    slots, total = [None] * world, [None]
    CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)

    def make_cb(rank):
        def cb(user, buf, count, stream):
            hip.hipStreamSynchronize(stream)
            host = np.empty(count, dtype=np.float64)
            hip.hipMemcpy(host.ctypes.data, buf, count * 8, 2)            # device -> host
            slots[rank] = host
            barrier.wait()
            if rank == 0:
                acc = slots[0].copy()
                for r in range(1, world):
                    acc += slots[r]
                total[0] = acc
            barrier.wait()
            hip.hipMemcpy(buf, total[0].ctypes.data, count * 8, 1)       # host -> device
            barrier.wait()
            return 0
        return CB(cb)
    outs, errs = [None] * world, []

    def run(rank):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                outs[rank] = dist_pvgo.run_chain_sharded(None, *args, LW, rank=rank, world=world, allreduce_cb=cbs[rank], params=params,
                                                            reproj=reproj)
        except Exception as e:                       # noqa: BLE001
            errs.append(e)
            barrier.abort()
    lib()
    cbs = [make_cb(r) for r in range(world)]
    torch.cuda.synchronize()
    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errs, errs
    return outs


@pytest.mark.parametrize('world,F', [(2, 300), (3, 257), (8, 5001), (5, 1000)])
def test_ranks_as_threads_through_the_allreduce_callback(cuda, world, F):
    """Exercises what the stage-level virtual-rank test cannot: message / halo / decision / final assembly of the C loop with
    more than one rank."""
    args = _problem(F, cuda)
    nodes, vels, res = _single(args)
    outs = _run_ranks_as_threads(args, world)
    for r, (n, v, rr, xb) in enumerate(outs):
        assert (rr.trials, rr.steps, rr.status) == (res.trials, res.steps, 0), r
        torch.testing.assert_close(n, nodes, rtol=0, atol=1e-9)
        torch.testing.assert_close(v, vels, rtol=0, atol=1e-9)
    if F == 5001:
        # ONE all-reduce per solve: 64.6 KB of interface blocks + [sum r^2, sum JD.(2R+JD), failed pivots] + the two raw parts of every
        # cut node's diagonal (18 doubles per cut); the first solve has no trial, every trial carries the next solve
        per_solve = (351 * 23 + 3 + 18 * (world - 1)) * 8
        solves, rest = divmod(outs[0][3], per_solve)
        trials = outs[0][2].trials
        # (no reject on this graph: no solve was redone; the host runs one trial ahead, so a stop may cancel one enqueued chain
        # whose collective still ran)
        assert rest == 0 and trials + 1 <= solves <= trials + 2


def _noisy(F, seed, sig, cuda):
    """Dead-reckoning init perturbed hard enough that LM has to reject trials (tests/test_pvgo_gpu.py::_noisy_problem)."""
    from oracle import lie
    prob, _ = chain_problem(F)
    rng = np.random.default_rng(seed)
    n = prob['init_nodes'].copy()
    n[:, :3] += rng.normal(0, sig, (F, 3))
    n = lie.se3_mul(lie.se3_exp(np.concatenate([np.zeros((F, 3)), rng.normal(0, sig * 0.2, (F, 3))], 1)), n)
    prob = dict(prob, init_nodes=n)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=cuda)
    return [t(prob[k]) for k in ('init_nodes', 'init_vels', 'vo_motions', 'imu_drots', 'imu_dtrans', 'imu_dvels', 'dts')]


@pytest.mark.parametrize('world,F,seed,sig', [(1, 65, 8, 1.5), (2, 65, 8, 1.5), (3, 65, 8, 1.5), (2, 65, 2, 1.0), (4, 65, 2, 1.0), (2, 33, 1, 1.5),
                                              (1, 1000, 3, 0.8), (2, 1000, 3, 0.8), (3, 777, 9, 3.0), (4, 5001, 5, 0.5), (8, 5001, 6, 1.5)])
def test_rejected_trials_cancel_the_run_ahead_chain_on_every_rank(cuda, world, F, seed, sig):
    """Rejects (more damping on the same linearisation; the pre-enqueued chain -- kernels AND collectives -- cancelled by the
    epoch gate) and the reject limit: the decision is taken on the device of every rank from the same summed scalars, so every
    rank must follow the accept / reject sequence of the fused single-GPU loop, and a second run must not see stale state."""
    from islam_amd import dist_pvgo, ops
    args = _noisy(F, seed, sig, cuda)
    nodes, vels = args[0].clone(), args[1].clone()
    res, trace = ops.pvgo_run_chain(nodes, vels, *args[2:], ops.pvgo_default_params(LW, radius=1e4), trace_cap=256)
    # (F > 96: the reject-heavy graphs of tests/test_pvgo_gpu.py -- rejects, verdict-5 re-solves and damping changes on the fused sharded loop)
    assert res.trials > res.steps or F > 96                               # there were rejected trials
    for _ in range(2):
        outs = [dist_pvgo.run_chain_sharded(None, *args, LW)] if world == 1 else _run_ranks_as_threads(args, world)
        for r, (n, v, rr, xb) in enumerate(outs):
            assert (rr.trials, rr.steps, rr.status) == (res.trials, res.steps, 0), r
            assert rr.loss == pytest.approx(res.loss, rel=1e-9) and rr.damping == pytest.approx(res.damping, rel=1e-12)
            torch.testing.assert_close(n, nodes, rtol=0, atol=1e-8)
            torch.testing.assert_close(v, vels, rtol=0, atol=1e-8)


@pytest.mark.parametrize('world,F', [(1, 33), (2, 33), (1, 300), (2, 300), (3, 513)])
def test_failed_solve_breaks_the_step_on_every_rank(cuda, world, F):
    """A negative information scalar makes the factorisation fail on some rank: the failed-pivot flag travels in all-reduce #2,
    every rank reports ISLAM_ENOTPD, re-linearises the same iterate (PyPose keeps looping through the scheduler) and stops on
    the plateau counter without moving (tests/test_pvgo_gpu.py::test_lm_solver_failure_breaks_the_step_like_pypose)."""
    from islam_amd import dist_pvgo, ops
    args = _problem(F, cuda)

    def prm():
        p = ops.pvgo_default_params(LW, radius=1e4)
        for i, w in enumerate((1.0, -0.5, 100.0, 0.01)):
            p.w[i] = w
        return p
    n0, v0 = args[0].clone(), args[1].clone()
    ref, _ = ops.pvgo_run_chain(n0, v0, *args[2:], prm())
    assert (ref.status, ref.steps, ref.trials) == (-3, 3, 3)              # the single-GPU loop: StopOnPlateau(patience=3) ends it
    outs = [dist_pvgo.run_chain_sharded(None, *args, LW, params=prm())] if world == 1 else _run_ranks_as_threads(args, world, params=prm())
    for n, v, rr, xb in outs:
        assert rr.status == -3 and rr.steps == 3 and rr.trials == 3
        assert rr.loss == pytest.approx(ref.loss, rel=1e-9)
        torch.testing.assert_close(n, args[0], rtol=0, atol=0)
        torch.testing.assert_close(v, args[1], rtol=0, atol=0)
    good = [dist_pvgo.run_chain_sharded(None, *args, LW)] if world == 1 else _run_ranks_as_threads(args, world)      # reusable afterwards
    assert all(o[2].status == 0 for o in good)


@pytest.mark.parametrize('world,F,K,compat', [(1, 65, 48, True), (2, 65, 48, True), (3, 129, 40, False), (4, 300, 64, True)])
def test_reprojection_factor_in_the_sharded_loop(cuda, world, F, K, compat):
    """The sparse reprojection factor (pvgo.py:53-61) couples consecutive nodes only, so it shards with the chain: every rank
    reduces the keypoints of its own links (global link 0 keeps the reference's frozen first motion, on rank 0 only).  Same LM
    trajectory as islam_pvgo_run_chain_reproj on one GPU ("absent from the sharded loop")."""
    from islam_amd import dist_pvgo, ops
    from tests.helpers import reproj_inputs
    from oracle import reproj as orp
    prob, tr = chain_problem(F)
    T_IL = np.array([0.1, -0.05, 0.02, 0.5, -0.5, 0.5, -0.5])
    inp = reproj_inputs(tr, K, T_IL)
    ref = orp.SparseReprojection(**inp)
    P3, tgt = ref.point3d, ref.target                                   # (M, K, 3) camera-frame keypoints, (M, K, 2) pixel targets
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=cuda)
    st = ops.pvgo_reproj_struct(t(P3), t(tgt), [inp['fx'], inp['fy'], inp['cx'], inp['cy']], T_IL, (2.0 / K) ** 2, compat)
    args = [t(prob[k]) for k in ('init_nodes', 'init_vels', 'vo_motions', 'imu_drots', 'imu_dtrans', 'imu_dvels', 'dts')]
    nodes, vels = args[0].clone(), args[1].clone()
    res, _ = ops.pvgo_run_chain(nodes, vels, *args[2:], ops.pvgo_default_params(LW, radius=1e4), reproj=st)
    n0, v0 = args[0].clone(), args[1].clone()
    plain, _ = ops.pvgo_run_chain(n0, v0, *args[2:], ops.pvgo_default_params(LW, radius=1e4))
    assert float((n0 - nodes).abs().max()) > 1e-4                        # the factor matters
    outs = [dist_pvgo.run_chain_sharded(None, *args, LW, reproj=st)] if world == 1 else _run_ranks_as_threads(args, world, reproj=st)
    for n, v, rr, xb in outs:
        assert (rr.trials, rr.steps, rr.status) == (res.trials, res.steps, 0)
        assert rr.loss == pytest.approx(res.loss, rel=1e-9)
        torch.testing.assert_close(n, nodes, rtol=0, atol=1e-9)
        torch.testing.assert_close(v, vels, rtol=0, atol=1e-9)
