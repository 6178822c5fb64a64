#!/usr/bin/env python3
"""Headline benchmark: PVGO Levenberg-Marquardt iterations per second on the 5000-frame graph.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.
For N>1 the driver launches it under torch.distributed.run (one rank per GPU, RCCL).

Workload = BASELINE.json configs[3] geometry on ONE graph ("Synthetic 5000-frame KITTI-shape graph"),
which is what BASELINE.json's metric is quoted on: N=5001 nodes / 5000 links, fp64, KITTI loss weights
(run_kitti.sh:5), IMU quantities produced by the HIP pre-integration kernels, everything resident in HBM
before the timed region.  One bench "step" = one complete run_pvgo optimisation of that graph from the same
initial state (pvgo.py:168-180: <=10 optimizer.step() calls, each with its inner damped retries).
An "LM iteration" = one damped normal-equation solve + trial step + loss / trust-region evaluation (one pass
of PyPose's inner ``while self.last <= self.loss`` loop); the linearisation of each optimizer.step() is
inside the timed region and amortised over its iterations.

value = LM iterations of all ranks' work / wall time (max over ranks).  N>1: the single graph is sharded
over the ranks (strong scaling; islam_amd/dist_pvgo.py), interface blocks exchanged over RCCL.
"""
import argparse
import json
import os
import sys
import time

# HIP deals its streams onto this many hardware queues in creation order (default 4).  The software-pipelined stereo_vio step runs the
# main chain, the prefetch stream and the fork streams of two captured graph copies: with four queues one fork stream shares the main
# stream's queue and the two serialise (494 instead of 880 frames/s).  Read when the HIP runtime initialises: set before torch is imported.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LOSS_WEIGHT = (1, 0.1, 10, 0.1)  # run_kitti.sh:5
N_FRAMES = 5001
# algorithmic HBM bytes per node of the launched level-0 elimination (bt_eliminate_tw_kernel<1>, first LM iteration and
# fallback solves), DESIGN.md section 3.1: read Hd 81 + Ho 81 + rhs 9, write fac 252 + inv 9 + damped diagonal 9 doubles
ELIM_BYTES_PER_NODE = (81 + 81 + 9 + 252 + 9 + 9) * 8
# ... of the dominant kernel of the LM loop's steady state (trial_elim_kernel: trial + linearisation + level-0 elimination),
# read nodes 7 + vels 3 + dx 9 + poses 7 + drots 4 + dtrans 3 + dvels 3 + dts 1 + old lin 42,
# write trial iterate 7 + 3, lin 42, Hd 81 + Ho 81 + rhs 9 (the linearisation the fallback solves read), fac 252 + inv 9 doubles
FUSED_BYTES_PER_NODE = (7 + 3 + 9 + 7 + 4 + 3 + 3 + 1 + 42 + 7 + 3 + 42 + 81 + 81 + 9 + 252 + 9) * 8
PRODUCT_BYTES_PER_SEGMENT = 351 * 8       # separator blocks, Schur contributions, fill handed to level 1


def build_problem(device, n_frames=N_FRAMES):
    """Config-4 graph; IMU dead-reckoning init and motion-mode deltas from the HIP integrator (train.py:236-246)."""
    from islam_amd import ops, synthetic
    tr = synthetic.car_trajectory(n_frames)
    t64 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=device)
    seg_h = np.ascontiguousarray(tr['rgb2imu_sync'] - tr['rgb2imu_sync'][0], dtype=np.int64)
    seg_d = torch.tensor(seg_h, device=device)
    dt, gyro, acc = t64(tr['imu_dts']), t64(tr['gyros']), t64(tr['accels'])
    ip, ir, iv = t64(tr['init']['pos']), t64(tr['init']['rot']), t64(tr['init']['vel'])
    z3 = torch.zeros(3, dtype=torch.float64, device=device)
    pos, rot, vel = ops.imu_preint(dt, gyro, acc, seg_d, seg_h, ip, ir, iv, tr['gravity'], False)
    dpos, drot, dvel = ops.imu_preint(dt, gyro, acc, seg_d, seg_h, z3, ir, z3, tr['gravity'], True)
    prob = dict(init_nodes=torch.cat([pos, rot], 1).contiguous(), init_vels=vel.contiguous(),
                vo=t64(tr['vo_motions']), drots=drot.contiguous(), dtrans=dpos.contiguous(), dvels=dvel.contiguous(),
                dts=t64(tr['dts']))
    return prob, tr


def perturbed_start(prob, sig, seed):
    """Dead-reckoning start perturbed by N(0, sig) m per axis in translation and N(0, 0.2 sig) rad per axis in rotation (what
    tests/test_pvgo_gpu.py::test_fused_loop_equals_the_launch_per_stage_loop_on_reject_heavy_graphs perturbs with the oracle's Lie algebra;
    here with the product's)."""
    from islam_amd import lietensor as pp
    g = torch.Generator().manual_seed(seed)
    N = prob['init_nodes'].shape[0]
    dt_ = torch.randn(N, 3, generator=g, dtype=torch.float64) * sig
    dr_ = torch.randn(N, 3, generator=g, dtype=torch.float64) * (0.2 * sig)
    n0 = prob['init_nodes'].cpu()
    n0 = torch.cat([n0[:, :3] + dt_, n0[:, 3:]], 1)
    pert = pp.SE3(torch.cat([torch.zeros(N, 3, dtype=torch.float64), pp.so3(dr_).Exp().tensor()], 1))
    return (pert @ pp.SE3(n0)).tensor().to(prob['init_nodes'].device).contiguous()


def lm_variant_rate(ops, prob, prm, device, start, what, runs=12):
    """LM rate of the full loop from `start` under `prm` (every run from the same state): iters/s, trials, accepted steps, rejected trials
    and damping changes of one run (its trace) -- the trials whose speculated damping missed are the launches behind them re-done."""
    N = prob['init_nodes'].shape[0]
    ws = ops.pvgo_workspace(N, device)
    states = [(start.clone(), prob['init_vels'].clone()) for _ in range(runs + 2)]
    trials = steps = 0
    trace = None
    for i, (nodes, vels) in enumerate(states):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        res, tr_ = ops.pvgo_run_chain(nodes, vels, prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'], prm, workspace=ws,
                                      trace_cap=256 if i == 0 else 0)
        if i == 0:
            trace = np.asarray(tr_)[:res.trials]
        if i >= 2:
            trials += res.trials
            steps += res.steps
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rej = int((trace[:, 2] == 0).sum()) if trace is not None and len(trace) else 0
    damp_changes = int((np.abs(np.diff(trace[:, 1])) > 0).sum()) if trace is not None and len(trace) > 1 else 0
    return {'value': trials / el, 'unit': 'LM iters/s', 'us_per_lm_iter': el / trials * 1e6, 'lm_iters_per_run': trials / runs,
            'accepted_steps_per_run': steps / runs, 'rejected_trials_per_run': rej, 'damping_changes_per_run': damp_changes,
            'accept_reject_pattern': ''.join('0' if a else '1' for a in trace[:, 2]) if trace is not None else None, 'what': what}


def reject_heavy_rate(ops, prob, device, radius=1e8):
    """The LM rate on a run that really REJECTS trials (the perturbed-start variant only moved the damping).
    Found with the oracle on the CPU (scripts/debug/reject_seed_search.py and the notes in DESIGN.md): on this graph no perturbation of the
    start makes LM reject -- Gauss-Newton steps on a 5000-link chain keep decreasing the loss, and small graphs reject only at their
    convergence floor, which this graph does not reach in ten steps -- but an over-optimistic initial trust region does: with
    TrustRegion(radius=1e8) (damping 1e-8 instead of pvgo.py:169's 1e-4) the first step's trial is rejected four times (damping 2e-8 ->
    8e-8 -> 6.4e-7 -> 1.0e-5) before it is accepted: oracle pattern 11110000000000, asserted against the HIP loop in
    tests/test_pvgo_gpu.py::test_rejecting_bench_configuration_matches_oracle.  Every rejected trial is an undo, a cancelled run-ahead
    iteration and a fallback solve from the stored linearisation under the new damping."""
    prm = ops.pvgo_default_params(LOSS_WEIGHT, radius=radius)
    return lm_variant_rate(ops, prob, prm, device, prob['init_nodes'],
                           'same graph and start, TrustRegion(radius=%g): the first optimizer.step rejects its trial four times before the damping has '
                           'grown enough (undo + cancelled run-ahead + fallback solve each time); counts and pattern (1 = rejected) are those of one run' % radius)


def trust_region_moving_rate(ops, prob, prm, device, sig=1.5, seed=6):
    """(`reject_heavy` of rounds 3-5) Same graph, perturbed start: every trial is accepted but the trust region moves, so trial_elim_kernel
    mis-speculates the damping and the host re-does the level-0 elimination from the stored linearisation."""
    return lm_variant_rate(ops, prob, prm, device, perturbed_start(prob, sig, seed),
                           'same graph, start perturbed by N(0, %.1f m) / N(0, %.2f rad) per axis (seed %d): no trial is rejected, the damping changes '
                           '-- trial_elim_kernel mis-speculates it and the host re-does the level-0 elimination; counts are those of one run' % (sig, 0.2 * sig, seed))


def eliminate_l0_burst(ops, Hd, Ho, rhs, N, levels, device, launches=40, seg_len=(0, 0), damping=0.0):
    """Average period (us) of back-to-back launches of the level-0 up-sweep kernel (all segments; the launch
    islam_pvgo_solve_chain makes, through islam_pvgo_eliminate_level0), HIP events on the stream the kernel is launched on
    (torch's current stream, the one islam_amd passes to the C ABI)."""
    from islam_amd._lib import c_double, c_int, c_size_t, check, lib, ptr, stream_ptr
    ws, nbytes = ops.pvgo_workspace(N, device)
    sl = (c_int * 2)(int(seg_len[0]), int(seg_len[1]))
    Hd = Hd.clone()

    def launch():
        check(lib().islam_pvgo_eliminate_level0(ptr(Hd), ptr(Ho), ptr(rhs), c_double(damping), N, sl, ptr(ws), c_size_t(nbytes),
                                                stream_ptr(device)))
    for _ in range(5):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(launches):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / launches


def latency_model(ops, Hd, Ho, rhs, levels, device, us_iter):
    """roofline.latency_model ("next to the byte roofline the latency model t >= n_launch * t_launch + depth * t_block"; VERDICT
    round 5, weak item 2).  Every term is measured live and in isolation, none is fitted to the LM loop it bounds:
      n_launch      dependent kernel launches of one steady-state LM iteration = trial_elim_kernel (trial + linearisation + level-0
                    elimination) + one bt_eliminate_tw_kernel per middle level + bt_downsweep_kernel (root + whole back-substitution)
      t_launch_us   period of a do-nothing kernel in a chain of dependent launches (islam_launch_cost_probe, 256 x 192 threads)
      depth_pivots  dependent 9x9 block-pivot node steps of the up-sweep along the critical path: a two-sided segment of m interior nodes
                    costs m // 2 + 1 steps, summed over the levels (the root's n nodes likewise)
      t_pivot_us    one such node step: slope of the isolated level-0 launch over the segment lengths 3 / 5 / 7 (2 / 3 / 4 steps) on a
                    1001-node prefix of the same normal equations -- at most 251 segments, at most one workgroup per CU
    NOT in the bound: the back-substitution's node steps (depth_backsub_steps, same count), the hand-offs between levels inside the
    down-sweep launch, the trial / linearisation work in front of the level-0 elimination -- bound_us is a floor, not an estimate."""
    import ctypes
    from islam_amd._lib import c_float, check, lib, stream_ptr
    us = c_float(0.0)
    check(lib().islam_launch_cost_probe(256, 192, 400, ctypes.byref(us), stream_ptr(device)))
    t_launch = float(us.value)
    n_probe = min(1001, Hd.shape[0])
    Hp, Op, rp = Hd[:n_probe].clone(), Ho[:n_probe].clone(), rhs[:n_probe].clone()
    steps_of = lambda m: m // 2 + 1 if m >= 3 else m
    pts = []
    for m in (3, 5, 7):
        pts.append((steps_of(m), eliminate_l0_burst(ops, Hp, Op, rp, n_probe, None, device, seg_len=(m, 0), damping=1e-4)))
    xs, ys = np.array([p[0] for p in pts], float), np.array([p[1] for p in pts], float)
    t_pivot = float(np.polyfit(xs, ys, 1)[0])
    depth = sum(steps_of(m) for (_, m, _) in levels)
    n_launch = len(levels)
    bound = n_launch * t_launch + depth * t_pivot
    return {'n_launch': n_launch, 't_launch_us': t_launch, 'depth_pivots': depth, 't_pivot_us': t_pivot, 'bound_us': bound,
            'achieved_us': us_iter, 'achieved_over_bound': us_iter / bound, 'depth_backsub_steps': depth,
            'level0_launch_us_by_segment_len': {str(m): float(y) for m, y in zip((3, 5, 7), ys)},
            'what': 't >= n_launch * t_launch + depth_pivots * t_pivot: launches and up-sweep pivots only (back-substitution steps, '
                    'level hand-offs and the trial / linearisation are not in the floor); t_launch from islam_launch_cost_probe, t_pivot from '
                    'isolated level-0 launches at segment lengths 3 / 5 / 7 on a %d-node prefix' % n_probe}


def trial_elim_burst(ops, prob, prm, N, device, launches=40):
    """Average period (us) of back-to-back launches of trial_elim_kernel (islam_pvgo_trial_elim_burst: one pair of HIP events
    on the launch stream around the burst, inside the library) + (segment length, segments, workgroups); None if the fused loop
    does not cover this N."""
    import ctypes
    from islam_amd._lib import IslamHipError, c_float, c_int, c_size_t, check, lib, ptr, stream_ptr
    ws, nbytes = ops.pvgo_workspace(N, device)
    us, info = c_float(0.0), (c_int * 3)()
    try:
        check(lib().islam_pvgo_trial_elim_burst(ptr(prob['init_nodes']), ptr(prob['init_vels']), ptr(prob['vo']), ptr(prob['drots']),
                                                ptr(prob['dtrans']), ptr(prob['dvels']), ptr(prob['dts']), N, ctypes.byref(prm), ptr(ws),
                                                c_size_t(nbytes), launches, ctypes.byref(us), info, stream_ptr(device)))
    except IslamHipError:
        return None, None
    return float(us.value), list(info)


def pvgo_source_sha16():
    """sha256[:16] over the PVGO translation unit: pvgo.hip and the pvgo_*.inl parts it includes, in name order (what
    scripts/make_traffic_json.py stamps the counter passes with)."""
    import glob
    import hashlib
    d = os.path.join(ROOT, 'islam_amd', 'csrc')
    h = hashlib.sha256()
    for f in [os.path.join(d, 'pvgo.hip')] + sorted(glob.glob(os.path.join(d, 'pvgo_*.inl'))):
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def committed_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (the latest profiles/traffic_r*.json), valid only for the kernel
    source they were taken on: the file records pvgo_source_sha16() (pvgo.hip + its pvgo_*.inl parts); a different source -> no traffic figure (never a stale one)."""
    import glob
    import hashlib
    fs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'traffic_r[0-9]*.json')))
    if not fs:
        return None
    tpath = fs[-1]
    t = json.load(open(tpath))
    t['file'] = os.path.relpath(tpath, ROOT)
    sha = pvgo_source_sha16()
    if t.get('pvgo_hip_sha16') != sha:
        return {'stale': True, 'note': '%s was taken on pvgo.hip %s, this is %s: dropped' % (t['file'], t.get('pvgo_hip_sha16'), sha)}
    return t


def cpu_baseline(prob_host):
    """Oracle (CPU restatement of the PyPose LM, oracle/pvgo.py) on the host cores: banded mode at full size."""
    from oracle import pvgo as opvgo
    try:
        from threadpoolctl import threadpool_limits
    except Exception:          # pragma: no cover
        threadpool_limits = None
    ctx = threadpool_limits(limits=1) if threadpool_limits else None
    # bounded sample: the full LM loop on the full graph, repeated until ~10 s of CPU work have been timed
    t0 = time.perf_counter()
    trials, runs = 0, 0
    while True:
        out = opvgo.run_pvgo(**prob_host, loss_weight=LOSS_WEIGHT, mode='banded', return_optimizer=True)
        trials += len(out[5].trace)
        runs += 1
        dt = time.perf_counter() - t0
        if dt >= 10.0 or runs >= 200:
            break
    if ctx is not None:
        ctx.unregister() if hasattr(ctx, 'unregister') else None
    # the faithful dense formulation (what PyPose builds), SURVEY 8(d) protocol on a bounded sample
    dense = cpu_dense_protocol(prob_host)
    return {
        'value': trials / dt, 'unit': 'LM iters/s', 'cores': 1, 'kind': 'port',
        'sample': 'oracle/pvgo.py banded (block-tridiagonal) mode, the full N=%d graph, %d full LM loops = %d LM iterations in '
                  '%.1f s, 1 thread (more threads make the many tiny LAPACK calls slower).  The north_star target (>= 30x the CPU PyPose LM rate at 5000 frames) has NO measured '
                  'denominator: PyPose is not installable here and its dense formulation needs > 60 GB at N = 5001 (dense_pypose_style stops at N = 513 '
                  'and is not extrapolated); speedup_vs_cpu_port is against this banded port only' % (prob_host['init_nodes'].shape[0], runs, trials, dt),
        'dense_pypose_style': dense,
    }


def cpu_dense_protocol(prob_host):
    """SURVEY 8(d) CPU protocol for the faithful formulation (dense J / block_diag W / dense Cholesky as PyPose builds them,
    oracle/pvgo.py mode='dense'): fp32 like the reference (pvgo.py:157-160) AND fp64, all host cores, 1 warm-up + median of 5 first
    LM iterations per size up to the largest size the bounded sample affords.  N=5001 itself needs > 60 GB and minutes per
    iteration and is NOT extrapolated from these sizes."""
    from oracle import pvgo as opvgo
    out = {'cores': os.cpu_count(), 'what': 'dense PyPose-style LM iteration, 1 warm-up + median of 5 (first optimizer.step of the loop)'}
    for name, dt in (('f32', np.float32), ('f64', np.float64)):
        rows = []
        for n_small in (129, 257, 513):
            small = {k: (v[:n_small] if k in ('init_nodes', 'init_vels') else v[:n_small - 1]) for k, v in prob_host.items()}
            ts = []
            for rep in range(6):
                t1 = time.perf_counter()
                o = opvgo.run_pvgo(**small, loss_weight=LOSS_WEIGHT, mode='dense', max_steps=1, return_optimizer=True, dtype=dt)
                ts.append((time.perf_counter() - t1) / max(len(o[5].trace), 1))
            rows.append((n_small, float(np.median(ts[1:]))))
        (n1, t1_), (n2, t2_) = rows[-2], rows[-1]
        expo = float(np.log(t2_ / t1_) / np.log(n2 / n1))
        # No N = 5001 figure is derived from these: at N <= 513 the fitted exponent is ~1.7-2.1 (launch / memory-bound), the asymptote of
        # dense J^T W J + Cholesky is 3, so a power-law extrapolation over a factor 10 in N would flatter the dense path by an unknown factor.
        out[name] = {'iters_per_s': {str(n): 1.0 / t for n, t in rows}, 'largest_N': rows[-1][0], 'value': 1.0 / rows[-1][1],
                     'fit_exponent_at_largest_N': expo,
                     'N5001': 'not measured and not extrapolated: needs > 60 GB (SURVEY F7); fit exponent %.2f at N <= %d is below the cubic asymptote' % (expo, n2)}
    return out


def vio_frames_per_sec(device, batch=8, steps=64, warmup=3):   # (16 timed steps = 0.2 s gave +-4 % run to run; 64: +-0.6 %)
    """Secondary metric of BASELINE.json ("stereo-VIO frames/sec", configs[1] shapes): the bilevel loop body of
    train.py:200-299 -- TartanVO forward at 448x640 (bf16 stereo net, HIP correlation/warp/scale), 2x IMU integrate,
    run_pvgo on the 9-node window, one-step backward -- on synthetic stereo pairs, random-init weights."""
    from islam_amd import lietensor as pp, nets as nets_mod, ops as ops_mod, synthetic
    from islam_amd.TartanVO import TartanVO
    from islam_amd.bilevel import BilevelLoop
    from islam_amd.imu_integrator import IMUModule
    torch.manual_seed(0)
    # batches whose frozen forward is queued ahead (= captured graph copies).  With the main chain shortened by the fused pose algebra
    # (islam_amd/glue.py) the step is bound by the replays: two copies, replays queued back to back, 884 / 885 against 849 / 874 frames/s
    # with one batch ahead on one box (before the fused algebra: 830 / 832 against 825 / 827, no gain).  Two copies need the eight
    # hardware queues set at the top of this file: with HIP's default four the second copy's fork stream shares the main stream's queue
    # and the two serialise (494).  Should that happen anyway (the run below checks: pipelined rate < 1.3 x sequential), the
    # measurement falls back to one batch ahead and says so.
    depth = int(os.environ.get('ISLAM_PREFETCH_DEPTH', '2'))
    vo = TartanVO(correct_scale=False, fix_parts=("flow", "stereo"), use_kitti_coord=True, frozen_dtype=torch.bfloat16, flow_dtype=torch.bfloat16, host_glue=True,
                  miopen_find=True, pose_channels_last=True,
                  pose_dtype=torch.bfloat16 if os.environ.get('ISLAM_POSE_BF16') == '1' else None,   # measured: 330 vs 325 frames/s -- not worth the numerics

                  graph_frozen=os.environ.get('ISLAM_NO_GRAPH') != '1', graph_instances=depth,
                  graph_pose=False if os.environ.get('ISLAM_NO_GRAPH') == '1' else {'callables': True, 'accumulate': 'accumulate'}.get(os.environ.get('ISLAM_POSE_GRAPH'), 'hip'))
    vo.fused_glue = os.environ.get('ISLAM_FUSED_GLUE', '1') == '1'      # the pose algebra behind the networks as one autograd node (islam_amd/glue.py)
    with torch.no_grad():      # random weights predict garbage disparity: pin the stereo head to 10 px so the scale mask is non-empty
        vo.vonet.stereoNet.conv_c13.weight.zero_()
        vo.vonet.stereoNet.conv_c13.bias.fill_(0.8)
    n = (steps + warmup) * batch + 1
    tr = synthetic.car_trajectory(n, seed=3)
    imu = IMUModule(tr['accels'], tr['gyros'], tr['imu_dts'], np.zeros(3), np.zeros(3), tr['init'], tr['gravity'],
                    tr['rgb2imu_sync'], device=str(device), denoise_model_name=None, denoise_accel=True, denoise_gyro=False)
    loop = BilevelLoop(vo, imu, pp.identity_SE3(), tr['init'], batch_size=batch, device=str(device))
    samples = []
    for k in range(2):
        smp = synthetic.stereo_batch(batch, seed=50 + k)
        samples.append({kk: (v.to(device) if isinstance(v, torch.Tensor) and (kk.startswith('img') or kk == 'intrinsic') else v)
                        for kk, v in smp.items()})
    # ---- diagnostics that let a slow line name its own cause: GPU-side duration of every frozen
    # graph replay (event pair on the side stream it runs on) and of every step's main chain (event pair on the main stream), the
    # shader clock sampled on a third stream before / during / after, and whether MIOpen served the pose head from the pinned set
    from islam_amd import miopen_pin
    # The probe's stream is created LAZILY, behind every stream of the pipeline (side stream of the prefetch, fork stream of the captured
    # graph, capture stream of the pose head): HIP deals streams onto a few hardware queues in creation order, and a probe stream made
    # first moved the prefetch's side stream onto the main stream's queue -- the two then ran one after the other (measured: 838 -> 460
    # frames/s, pipelined == sequential).
    probe_box = []

    def probe():
        if not probe_box:
            probe_box.append(ops_mod.ClockProbe(device, capacity=96))
        return probe_box[0]
    replay_ev = []
    inner = vo.vonet._frozen_graphed

    diag_mode = os.environ.get('ISLAM_VIO_DIAG', 'all')       # all | none | replay | chain | probe  (A/B runs of the instrumentation itself)
    ev_replay, ev_chain, ev_probe = (diag_mode in ('all', 'replay')), (diag_mode in ('all', 'chain')), (diag_mode in ('all', 'probe'))

    def timed_replay(imgs):
        if not ev_replay:
            return inner(imgs)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = inner(imgs)
        b.record()
        replay_ev.append((a, b))
        return r
    vo.vonet._frozen_graphed = timed_replay

    def run(pipelined, probe_every=0, depth=depth):
        loop.reset()
        seq = []
        for k in range(steps + warmup + 3):
            smp = dict(samples[k % 2])
            smp['link'] = samples[k % 2]['link'] + k * batch
            seq.append(smp)
        t0 = 0.0
        chain_ev = []
        for k in range(steps + warmup):
            if k == warmup:
                torch.cuda.synchronize()
                loop.timing = dict(vo=0.0, imu=0.0, pgo=0.0, opt=0.0)
                del replay_ev[:]
                t0 = time.perf_counter()
            if ev_probe and probe_every and k >= warmup and (k - warmup) % probe_every == 0:
                probe().sample()
            if ev_chain:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            ahead = tuple(seq[k + 1:k + 1 + depth]) if depth > 1 else seq[k + 1]
            loop.step(seq[k], next_sample=ahead if pipelined else None)
            if ev_chain:
                b.record()
                if k >= warmup:
                    chain_ev.append((a, b))
        torch.cuda.synchronize()
        el_ = time.perf_counter() - t0
        med = lambda ev: float(np.median([x.elapsed_time(y) for x, y in ev])) if ev else None
        return el_, dict(loop.timing), {'frozen_replay_gpu_ms': med(replay_ev[-steps:]), 'main_chain_gpu_ms': med(chain_ev)}

    el_seq, tm, gpu_seq = run(False)
    # the frozen flow / disparity forward of batch k+1 overlaps the IMU / PVGO / backward of batch k.  THREE pipelined runs: the line
    # carries median / min / max (value = the median run)
    pipe = [run(True, probe_every=8)]
    fallback = None
    if depth > 1 and steps * batch / pipe[0][0] < 1.3 * (steps * batch / el_seq):
        # the deeper schedule did not overlap (a stream of the second graph copy sharing the main stream's hardware queue?): one batch ahead
        fallback = {'depth_%d_frames_per_s' % depth: steps * batch / pipe[0][0], 'now': 'one batch ahead'}
        depth = 1
        pipe = [run(True, probe_every=8, depth=1)]
    pipe += [run(True, probe_every=8, depth=depth) for _ in range(2)]
    clock_run = probe().n
    time.sleep(0.3)
    for _ in range(3):
        probe().sample()
    mhz = probe().mhz()
    order = sorted(range(3), key=lambda r: pipe[r][0])
    el, tm_pipe, gpu_pipe = pipe[order[1]]
    rates = [steps * batch / pipe[r][0] for r in range(3)]
    pin_ok, pin_msg = miopen_pin.check_pinned_db(device.index or 0, strict=False)
    diag = {
        'pipelined_runs_frames_per_s': {'median': float(np.median(rates)), 'min': min(rates), 'max': max(rates), 'runs': rates},
        'pipelined_host_stage_ms_per_batch': {k: v / steps * 1e3 for k, v in tm_pipe.items()},
        'gpu_side_ms_per_step': {
            'pipelined': gpu_pipe, 'sequential': gpu_seq,
            'what': 'HIP event pairs: frozen_replay = around the frozen nets\' graph replay on the stream it runs on (side stream when pipelined); '
                    'main_chain = around BilevelLoop.step on the main stream (pose head, glue, IMU, PVGO, backward; includes host gaps and, '
                    'sequentially, the replay itself); medians over the timed steps of the median run'},
        'shader_clock_mhz': {'during_pipelined_median': float(np.median(mhz[:clock_run])) if clock_run else None,
                             'during_pipelined_min': min(mhz[:clock_run]) if clock_run else None,
                             'idle_after': mhz[clock_run:],
                             'what': 'islam_clock_probe: one wavefront, dependent fp64 FMA chain, shader cycles / constant-rate wall clock, on a stream of its own'},
        'miopen_pinned_db': {'matches_device_and_version': bool(pin_ok), 'detail': pin_msg, 'searched_in_this_process': miopen_pin.searched_since_start()},
    }
    out = {'value': steps * batch / el, 'unit': 'frames/s', 'batch': batch, 'image': '448x640 stereo',
           'nets': nets_mod.execution_description() + (' | pose head fp32 (trainable): forward and backward on the hand-written exact-fp32 matrix-core kernels of csrc/pose_head.hip, each a HIP graph'
                              if vo.vonet.graph_pose == 'hip' else ' | pose head fp32 (trainable, MIOpen with a pinned solution set; forward / backward as HIP graphs)')
                   + ' | PVGO of the 9-node window: the whole LM loop in one launch (small_lm_kernel)',
           'gflop_per_frame': 466.4, 'tflops': 466.4e-3 * steps * batch / el, 'mfma_frac': 466.4e-3 * steps * batch / el / 2500.0,
           'ms_per_batch': el / steps * 1e3,
           'schedule': 'software-pipelined: TartanVO.prefetch queues the frozen nets of the next %d batch(es) on a side stream (%d captured graph copies, round-robin)' % (depth, depth),
           'schedule_fallback': fallback,
           'sequential_frames_per_s': steps * batch / el_seq, 'sequential_ms_per_batch': el_seq / steps * 1e3,
           'sequential_stage_ms_per_batch': {k: v / steps * 1e3 for k, v in tm.items()},
           'forward_only_frames_per_s': steps * batch / tm['vo'], 'weights': 'random init', 'data': 'synthetic',
           'diagnostics': diag}
    # roofline of the front end with EXECUTED matrix-core work (the reference-equivalent 466.4 GFLOP per frame above counts the
    # full-resolution tail the quarter-resolution evaluation skips) and the shader clock the chip sustains under this load: both from
    # the committed hardware-counter pass over the frozen forward (scripts/frozen_pmc.sh -> profiles/r*/frozen_exec_summary_*.json, latest round)
    ex = committed_exec_summary()
    if ex:
        g = ex['executed_gflop_per_frame']
        clk = ex.get('shader_clock_ghz_time_weighted') or 2.4
        rate = g * 1e-3 * steps * batch / el
        out['roofline'] = {'bound': 'mfma', 'executed_gflop_per_frame': g, 'achieved': rate, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': rate / 2500.0,
                           'shader_clock_ghz': clk, 'peak_at_clock': 2500.0 * clk / 2.4, 'frac_at_clock': rate / (2500.0 * clk / 2.4),
                           'frozen_kernel_ms_per_batch': ex.get('kernel_ms_per_forward'),
                           'source': ex.get('source'),
                           'note': 'frames/s x the matrix-pipe work the frozen nets EXECUTE per frame: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 '
                                   'SIMDs, one eager forward at B=8 under rocprofv3 --pmc) x 1024 bf16 flop per busy cycle / 8 frames.  It counts what '
                                   'the pipe did -- tile and channel padding included, the skipped full-resolution tail excluded -- which is why it '
                                   'differs from the reference-equivalent gflop_per_frame above; the trainable pose head (fp32, MIOpen) is not in it.  '
                                   'shader_clock_ghz: GRBM_GUI_ACTIVE / 8 XCDs / duration over the kernels of >= 50 us of that pass (one kernel at a '
                                   'time; the pipelined step runs two streams and may clock lower)'}
    return out


def vio_cpu_baseline(budget_s=15.0, window=8):
    """stereo_vio.cpu_baseline: SURVEY 8(d)'s frames/s definition --
    B / wall(tartanvo(sample) + 2 x integrate + run_pvgo) -- on the host cores, from the CPU restatements only: the fp32 eager network
    definitions of islam_amd/nets.py (the reference's architectures, torch CPU convolutions, train-mode BatchNorm like TartanVO.py:91)
    with the two native ops replaced by the oracle's C (oracle/corr81.c: correlation, warp), the oracle glue (oracle/tartanvo.py: Canny
    edge mask, stereo scale, frame change), oracle IMU pre-integration and the oracle's DENSE (PyPose-style) LM on the 9-node window.
    Bounded sample: 1 warm-up + as many B = 1 frames as fit ~budget_s, the window's IMU + PVGO timed once per `window` frames."""
    import platform
    from islam_amd import nets, synthetic
    from oracle import cwrap, imu as oimu, pvgo as opvgo, tartanvo as otvo
    # (B = 1 convolutions do not scale past a few dozen threads: all 256 threads of the GPU box's host ran a frame in 79 s, 8 threads of the
    #  build container in 17 s)
    cores = min(os.cpu_count() or 1, int(os.environ.get('ISLAM_CPU_BASELINE_THREADS', '32')))
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    saved = nets.corr_fn, nets.warp_fn
    nets.corr_fn = lambda a, b: torch.from_numpy(cwrap.corr81_fwd(a.numpy(), b.numpy()))
    nets.warp_fn = lambda x, f, sc: torch.from_numpy(cwrap.warp(x.numpy(), (f * sc).numpy()))
    try:
        torch.manual_seed(0)
        vn = nets.VONet(fix_parts=('flow', 'stereo')).train()
        with torch.no_grad():                                 # (as the GPU leg: pin the random-init stereo head to 10 px)
            vn.stereoNet.conv_c13.weight.zero_()
            vn.stereoNet.conv_c13.bias.fill_(0.8)
        smp = synthetic.stereo_batch(1, seed=50)
        args = [smp[k] for k in ('img0', 'img1', 'img0_norm', 'img0_r_norm', 'intrinsic')]
        calib, base = smp['intrinsic_calib'].numpy(), smp['extrinsic'][:, 0].numpy()

        def frame():
            t = time.perf_counter()
            with torch.no_grad():
                flow, disp, pose = vn(*args)
            t_net = time.perf_counter() - t
            otvo.forward_glue(flow.numpy(), disp.numpy(), pose.numpy(), smp['img0'].numpy(), calib, base, smp['datatype'])
            return t_net, time.perf_counter() - t
        t_all = time.perf_counter()
        warm = frame()                                        # warm-up (oneDNN primitive creation, first-touch of the buffers)
        nets_s, frames_s = [], []
        # (a host on which the warm-up frame alone exceeds the budget gets ONE timed frame: the default bench run must stay within minutes)
        min_frames = 1 if warm[1] > budget_s else 2
        t_all = time.perf_counter()
        while len(frames_s) < min_frames or (time.perf_counter() - t_all < budget_s and len(frames_s) < 64):
            a, b = frame()
            nets_s.append(a)
            frames_s.append(b)
        # the window's back end: 2 x integrate (world + motion rows, train.py:236-246) and run_pvgo on window + 1 nodes, dense like PyPose
        F = window + 1
        tr = synthetic.car_trajectory(F, seed=3)
        t = time.perf_counter()
        pos, rot, vel = oimu.integrate(tr['accels'], tr['gyros'], tr['imu_dts'], tr['rgb2imu_sync'], 0, F - 1, tr['init'], tr['gravity'], False)
        dpos, drot, dvel = oimu.integrate(tr['accels'], tr['gyros'], tr['imu_dts'], tr['rgb2imu_sync'], 0, F - 1, tr['init'], tr['gravity'], True)
        t_imu = time.perf_counter() - t
        prob = synthetic.pvgo_problem_from_deltas(tr, drot, dpos, dvel, pos, rot, vel)
        t = time.perf_counter()
        opt = opvgo.run_pvgo(**prob, loss_weight=LOSS_WEIGHT, mode='dense', return_optimizer=True)[5]
        t_pgo = time.perf_counter() - t
        per_frame = float(np.median(frames_s)) + (t_imu + t_pgo) / window
        model = platform.processor() or ''
        try:
            for line in open('/proc/cpuinfo'):
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
        except Exception:
            pass
        return {'value': 1.0 / per_frame, 'unit': 'frames/s', 'cores': cores, 'cpu': model, 'kind': 'port',
                'ms_per_frame': {'networks_fp32_eager': float(np.median(nets_s)) * 1e3, 'glue_edge_scale': (float(np.median(frames_s)) - float(np.median(nets_s))) * 1e3,
                                 'imu_2x_integrate_per_window': t_imu * 1e3, 'pvgo_dense_lm_per_window': t_pgo * 1e3, 'window_frames': window,
                                 'pvgo_lm_iters': len(opt.trace)},
                'sample': 'SURVEY 8(d) frames/s (forward + IMU + PVGO, no backward) at B = 1, 448x640: %d frames after 1 warm-up in %.1f s (median frame), '
                          'torch %s CPU convolutions on %d threads for the nets (islam_amd/nets.py fp32 eager definitions, train-mode BatchNorm), '
                          'oracle C correlation / warp, oracle glue, oracle IMU, oracle dense LM on the %d-node window amortised over %d frames; the GPU '
                          'figure beside it is B = 8 and also runs the backward + optimiser step'
                          % (len(frames_s), time.perf_counter() - t_all, torch.__version__, cores, F, window)}
    finally:
        nets.corr_fn, nets.warp_fn = saved
        torch.set_num_threads(prev_threads)


def committed_exec_summary():
    """profiles/r*/frozen_exec_summary_*.json (latest round): executed matrix-core GFLOP per frame and the shader clock of the frozen forward,
    from rocprofv3 PMC passes on the GPU box; None if no summary is committed."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9]*', 'frozen_exec_summary_*.json')))
    if not fs:
        return None
    try:
        d = json.load(open(fs[-1]))
        d['source'] = os.path.relpath(fs[-1], ROOT)
        return d
    except Exception:
        return None


def front_end_kernel_rooflines(device):
    """HBM rooflines of the two bandwidth-bound front-end kernels north_star names (81-channel correlation, warp + mask) at the PWC
    level where they move the most bytes (level 2: C = 32, 112 x 160, B = 8): a burst between one event pair on the launch stream,
    algorithmic bytes of SURVEY 8(d) (4 B H W (2 C + 81) and 4 B H W (2 C + 2))."""
    from islam_amd import ops
    B, C, H, W = 8, 32, 112, 160
    g = torch.Generator().manual_seed(3)
    f1 = torch.randn(B, C, H, W, generator=g).to(device)
    f2 = torch.randn(B, C, H, W, generator=g).to(device)
    # a smooth field of a few pixels (what an optical-flow estimate looks like), not per-pixel noise: neighbouring lanes gather neighbouring texels
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    flow = torch.stack([0.6 * torch.sin(yy / 17.0) + 0.3 * torch.cos(xx / 23.0), 0.5 * torch.cos(yy / 29.0 + xx / 31.0)], 0)
    flow = (flow[None].repeat(B, 1, 1, 1) + 0.02 * torch.randn(B, 2, H, W, generator=g)).contiguous().to(device)
    buf = torch.empty(B, 81 + 8, H, W, device=device)

    def burst(fn, n=40):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    out = {}
    us = burst(lambda: ops.corr81_act(f1, f2, buf, 8, 0.1))
    by = 4.0 * B * H * W * (2 * C + 81)
    out['corr81_fwd_level2'] = {'us': us, 'bytes': by, 'GB/s': by / us / 1e3, 'frac': by / us / 1e3 / HBM_PEAK_GBS, 'timing': 'burst of launches, one event pair'}
    us = burst(lambda: ops.warp_mask(f2, flow, 5.0))
    by = 4.0 * B * H * W * (2 * C + 2)
    out['warp_mask_level2'] = {'us': us, 'bytes': by, 'GB/s': by / us / 1e3, 'frac': by / us / 1e3 / HBM_PEAK_GBS, 'timing': 'burst of launches, one event pair'}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)       # one step = one full run_pvgo LM loop (~0.8 ms on one MI355X)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--frames', type=int, default=N_FRAMES)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-frontend', action='store_true')
    args = ap.parse_args()

    # stdout carries ONE JSON line and nothing else: libraries that print banners to file descriptor 1 while they initialise (RCCL's
    # version / host block on rank 0) are pointed at stderr until the line is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d'
                             % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    dev_index = local_rank % torch.cuda.device_count()     # (several ranks share a GPU only in the gloo self-test below)
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    dist = None
    force_sharded = os.environ.get('ISLAM_FORCE_SHARDED') == '1'   # 1-GPU self-test of the N>1 code path (RCCL, world 1)
    if world > 1 or force_sharded:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        backend = os.environ.get('ISLAM_DIST_BACKEND', 'nccl')      # 'nccl' is RCCL on ROCm; 'gloo' only for 1-GPU self-tests
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    from islam_amd import ops
    from islam_amd.miopen_pin import use_pinned_db
    use_pinned_db()          # before the process's first convolution: the front-end's MIOpen kernels are pinned (islam_amd/miopen_pin.py)
    prob, tr = build_problem(device, args.frames)
    N = prob['init_nodes'].shape[0]
    prm = ops.pvgo_default_params(LOSS_WEIGHT, radius=1e4)

    # The timed region is PASSES x (exactly --steps steps): one pass is the contract's K steps between barrier + synchronize on both sides
    # (max over ranks); with the driver's --steps 20 one pass is ~11 ms, where one scheduler hiccup moves the figure by percent, so the
    # K steps are timed `passes` times back to back (>= 0.1 s in total) and the MEDIAN pass is reported -- steps / ms_per_step keep their
    # meaning.
    passes = max(1, int(os.environ.get('ISLAM_BENCH_PASSES', str(-(-200 // max(args.steps, 1))))))
    rccl_ranks_seen = None
    if world == 1 and not force_sharded:
        ws = ops.pvgo_workspace(N, device)
        # run_pvgo works IN PLACE on the iterate, so every step gets its own copy of the initial state, resident in HBM before
        # the timed region starts (400 KB per step); the timed loop holds nothing but run_pvgo calls
        states = [(prob['init_nodes'].clone(), prob['init_vels'].clone()) for _ in range(args.warmup + passes * args.steps)]
        state_iter = iter(states)

        def step():
            nodes, vels = next(state_iter)
            res, _ = ops.pvgo_run_chain(nodes, vels, prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'],
                                        prm, workspace=ws)
            return res.trials, res.steps
    else:
        # N > 1: ONE graph sharded over the ranks, the whole LM loop inside the library on RCCL (islam_pvgo_run_chain_sharded,
        # csrc/pvgo_dist.hip -> pvgo.hip: run_chain_sharded_fused): per LM trial ONE all-reduce -- the interface blocks of the next solve,
        # the trial's loss / trust-region scalars and the cut nodes' diagonal parts.  ISLAM_SHARDED_PYTHON=1: the Python-driven stage
        # loop (islam_amd/dist_pvgo.py: two all-reduces per trial).
        from islam_amd import dist_pvgo
        sharded_info = {}
        if os.environ.get('ISLAM_SHARDED_PYTHON') == '1':
            solver = dist_pvgo.ShardedChainPVGO(prob['init_nodes'], prob['init_vels'], prob['vo'], prob['drots'], prob['dtrans'],
                                                prob['dvels'], prob['dts'], LOSS_WEIGHT, radius=1e4, group=None)

            def step():
                r = solver.run()
                return r['trials'], r['steps']
            sharded_info['loop'] = 'python stage calls + torch.distributed'
        else:
            comm = dist_pvgo.RcclComm(device=device)
            # what RCCL itself reports for the communicator the collectives run on (ncclCommCount): a SCALE record with
            # rccl_ranks_seen == n_gpus proves the all-reduces spanned that many ranks
            rccl_ranks_seen = comm.info()[0]
            sharded_info['rccl'] = dict(zip(('ranks', 'rank', 'device'), comm.info()))

            def step():
                _, _, res, xb = dist_pvgo.run_chain_sharded(comm, prob['init_nodes'], prob['init_vels'], prob['vo'], prob['drots'],
                                                            prob['dtrans'], prob['dvels'], prob['dts'], LOSS_WEIGHT, radius=1e4)
                sharded_info['exchange_bytes_per_lm_iter'] = xb / max(res.trials, 1)
                return res.trials, res.steps
            sharded_info['loop'] = 'islam_pvgo_run_chain_sharded (C, RCCL): fused trial + level-0 elimination kernel, one all-reduce per LM trial'

    for _ in range(args.warmup):
        step()
    pass_s, trials, steps_lm = [], 0, 0
    for _ in range(passes):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        tr_p = st_p = 0
        for _ in range(args.steps):
            a, b = step()
            tr_p += a
            st_p += b
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el_p = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el_p], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_p = t.item()
        pass_s.append(el_p)
        trials, steps_lm = tr_p, st_p            # (every pass does the same work from the same initial states)
    elapsed = float(np.median(pass_s))

    # ---- a second strong-scaling point on a graph LARGE enough for sharding to pay: the 5000-frame
    # graph of the headline is one 57 us chain of dependent launches per LM iteration -- a latency-bound all-reduce per trial costs
    # about as much -- while at N = 300 007 one iteration is 1.3 ms of work on one GPU.  Same code path as the headline at every world
    # size (fused single-GPU loop at world 1, islam_pvgo_run_chain_sharded otherwise); ISLAM_BENCH_LARGE_N=0 skips it.
    large = None
    big_n = int(os.environ.get('ISLAM_BENCH_LARGE_N', '300007'))
    if big_n > 0:
        try:
            t_b = time.perf_counter()
            ok_b = 1.0
            try:
                prob_b, _ = build_problem(device, big_n)
            except Exception:
                if dist is None:
                    raise
                ok_b = 0.0
            if dist is not None:
                # every rank must enter the collectives below or none: a rank-local failure (out of memory while building the 300k
                # problem) is agreed on FIRST -- otherwise the other ranks would wait in an all-reduce this rank never joins
                okt = torch.tensor([ok_b], dtype=torch.float64, device=device)
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                if okt.item() < 1.0:
                    raise RuntimeError('large-graph section skipped on every rank: building the N=%d problem failed on at least one' % big_n)
            build_s = time.perf_counter() - t_b
            runs_b = 3
            if world == 1 and not force_sharded:
                ws_b = ops.pvgo_workspace(big_n, device)
                st_b = [(prob_b['init_nodes'].clone(), prob_b['init_vels'].clone()) for _ in range(runs_b + 1)]

                def step_b(i):
                    r, _ = ops.pvgo_run_chain(st_b[i][0], st_b[i][1], prob_b['vo'], prob_b['drots'], prob_b['dtrans'], prob_b['dvels'], prob_b['dts'],
                                              prm, workspace=ws_b)
                    return r.trials
            elif os.environ.get('ISLAM_SHARDED_PYTHON') == '1':       # (the gloo self-test of the plumbing: Python stage loop)
                solver_b = dist_pvgo.ShardedChainPVGO(prob_b['init_nodes'], prob_b['init_vels'], prob_b['vo'], prob_b['drots'], prob_b['dtrans'],
                                                      prob_b['dvels'], prob_b['dts'], LOSS_WEIGHT, radius=1e4, group=None)

                def step_b(i):
                    return solver_b.run()['trials']
            else:
                def step_b(i):
                    _, _, r, _ = dist_pvgo.run_chain_sharded(comm, prob_b['init_nodes'], prob_b['init_vels'], prob_b['vo'], prob_b['drots'],
                                                             prob_b['dtrans'], prob_b['dvels'], prob_b['dts'], LOSS_WEIGHT, radius=1e4)
                    return r.trials
            # (behind the agreed set-up a failure inside the timed collectives is not caught for world > 1: the launcher tears the job down)
            step_b(0)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t_b = time.perf_counter()
            tr_b = sum(step_b(1 + i) for i in range(runs_b))
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            el_b = time.perf_counter() - t_b
            if dist is not None:
                tb_ = torch.tensor([el_b], dtype=torch.float64, device=device)
                dist.all_reduce(tb_, op=dist.ReduceOp.MAX)
                el_b = tb_.item()
            large = {'N': big_n, 'value': tr_b / el_b, 'unit': 'LM iters/s', 'us_per_lm_iter': el_b / tr_b * 1e6, 'n_gpus': world, 'scaling': 'strong',
                     'runs': runs_b, 'lm_iters_per_run': tr_b / runs_b, 'problem_build_s': build_s,
                     'what': 'the same chain graph at N = %d nodes, one graph over %d GPU(s): the size at which sharding can pay' % (big_n, world)}
            del prob_b
        except Exception as e:           # the headline metric must still be reported
            if dist is not None and 'skipped on every rank' not in str(e):
                raise                    # world > 1: the other ranks are inside collectives this rank left -- fail the job, do not hang it
            large = {'error': repr(e)[:300]}

    # ---- N > 1 (and the 1-GPU self-test of that path): what a SCALE record is checked against, line by line
    # (profiles/r05/scaling_model_r05.json: t(P) = t_fused(N / P) + t_shard + t_allreduce).  Every rank times the FUSED single-GPU loop on
    # a chain of its own stretch's length (N / P nodes, same generator): t_fused(N / P) measured where the sharded run ran.
    if dist is not None:
        n_loc = (N - 1) // world + 1
        prob_l, _ = build_problem(device, n_loc)
        ws_l = ops.pvgo_workspace(n_loc, device)
        st_l = [(prob_l['init_nodes'].clone(), prob_l['init_vels'].clone()) for _ in range(6)]
        tr_l = 0
        for i, (n_, v_) in enumerate(st_l):
            if i == 1:
                torch.cuda.synchronize()
                t_l = time.perf_counter()
            r_l, _ = ops.pvgo_run_chain(n_, v_, prob_l['vo'], prob_l['drots'], prob_l['dtrans'], prob_l['dvels'], prob_l['dts'], prm, workspace=ws_l)
            if i >= 1:
                tr_l += r_l.trials
        torch.cuda.synchronize()
        t_loc = torch.tensor([(time.perf_counter() - t_l) / max(tr_l, 1) * 1e6], dtype=torch.float64, device=device)
        t_all = [torch.zeros_like(t_loc) for _ in range(world)]
        dist.all_gather(t_all, t_loc)
        sharded_info['nodes_per_rank'] = n_loc
        sharded_info['t_fused_us_per_lm_iter_per_rank'] = [float(t.item()) for t in t_all]
        sharded_info['t_fused_what'] = 'the fused single-GPU LM loop on a chain of N / P nodes, timed on every rank (5 runs): the first term of the scaling model'
        del prob_l, ws_l, st_l

    # ---- N > 1 only: the same graph solved independently on every GPU (one trajectory per GPU, SURVEY 8(e) row 4: no
    # data-path collective), reported NEXT to the headline sharded-graph figure, never instead of it
    replicas = None
    if dist is not None:
        # (no try / except: collectives inside -- a rank that dropped out would leave the others waiting; the launcher tears the job down)
        ws_r = ops.pvgo_workspace(N, device)
        n_r, v_r = torch.empty_like(prob['init_nodes']), torch.empty_like(prob['init_vels'])

        def step_r():
            n_r.copy_(prob['init_nodes'])
            v_r.copy_(prob['init_vels'])
            res, _ = ops.pvgo_run_chain(n_r, v_r, prob['vo'], prob['drots'], prob['dtrans'], prob['dvels'], prob['dts'],
                                        prm, workspace=ws_r)
            return res.trials
        for _ in range(args.warmup):
            step_r()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        tr_r = 0
        for _ in range(args.steps):
            tr_r += step_r()
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0, float(tr_r)], dtype=torch.float64, device=device)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        replicas = {'value': t[1].item() / tmax[0].item(), 'unit': 'LM iters/s', 'scaling': 'weak',
                    'what': 'every rank runs the fused single-GPU LM loop on its own copy of the N=%d graph '
                            '(independent trajectories, no collective); all ranks\' iterations / max-over-ranks time' % N}

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream
        lin, _ = ops.pvgo_linearize(prob['init_nodes'], prob['init_vels'], prob['vo'], prob['drots'], prob['dtrans'],
                                    prob['dvels'], prob['dts'])
        w4 = [x ** 2 for x in LOSS_WEIGHT]
        Hd, Ho, rhs = ops.pvgo_build_normal(lin, prob['dts'], N, w4)
        ws2 = ops.pvgo_workspace(N, device)
        reps, acc_ms, levels = 30, None, None
        for i in range(reps + 3):
            _, ms, levels = ops.pvgo_solve_chain_timed(Hd.clone(), Ho, rhs, 1e-4, workspace=ws2)
            if i >= 3:
                acc_ms = dict(ms) if acc_ms is None else {k: acc_ms[k] + ms[k] for k in ms}
        kern = {k: round(v / reps * 1e3, 2) for k, v in acc_ms.items()}      # microseconds per launch (one event pair each)
        # the dominant kernel alone: a burst of back-to-back launches between ONE pair of events on the launch stream
        # (an event after every launch, as in the per-level table above, adds ~2 us to each)
        elim0_s = eliminate_l0_burst(ops, Hd, Ho, rhs, N, levels, device) * 1e-6
        kern['eliminate_L0_burst'] = round(elim0_s * 1e6, 2)
        elim_bytes = ELIM_BYTES_PER_NODE * N + PRODUCT_BYTES_PER_SEGMENT * levels[0][2]     # + the per-segment products handed to level 1
        # the dominant kernel of the LM loop's steady state: trial + linearisation + level-0 elimination in one launch
        fused_us, fused_info = trial_elim_burst(ops, prob, prm, N, device)
        tfile = committed_traffic() if N == N_FRAMES else None         # HBM bytes per launch from committed rocprofv3 PMC passes
        tk = (tfile or {}).get('kernels', {}) if tfile and not tfile.get('stale') else {}
        if fused_us:
            dom_name, dom_s = 'trial_elim_kernel (trial + linearisation + level-0 elimination)', fused_us * 1e-6
            alg_bytes = FUSED_BYTES_PER_NODE * N + PRODUCT_BYTES_PER_SEGMENT * levels[0][2]
            traffic = tk.get('trial_elim_kernel', {}).get('traffic_bytes_per_launch')
        else:                                                            # the fused loop does not cover this N
            dom_name, dom_s, alg_bytes = 'bt_eliminate_tw_kernel (level 0)', elim0_s, elim_bytes
            traffic = tk.get('bt_eliminate_tw_kernel_L0', {}).get('traffic_bytes_per_launch')
        achieved = alg_bytes / dom_s / 1e9
        # a burst of whole solves (all launches back to back, one event pair): what the solve costs inside the LM loop
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        Hb = Hd.clone()
        for _ in range(3):
            ops.pvgo_solve_chain_enqueue(Hb, Ho, rhs, ws2, damping=1e-4)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(40):
            ops.pvgo_solve_chain_enqueue(Hb, Ho, rhs, ws2, damping=1e-4)    # (the un-anchored graph is singular without damping)
        e1.record()
        torch.cuda.synchronize()
        ops.pvgo_solve_status(N, ws2, device)
        solve_us = e0.elapsed_time(e1) * 1e3 / 40
        us_iter = elapsed / trials * 1e6
        n_all = sum(l[0] for l in levels)                               # nodes of every level of the tree
        # algorithmic bytes of the other two launch families of an LM iteration
        sweep_bytes = (252 + 9 + 9) * 8 * n_all                         # factor rows + reciprocal pivots read, solution written
        sweep_key = [k for k in kern if 'downsweep' in k or k.startswith('top')]
        sweep_us = kern[sweep_key[0]] if sweep_key else None
        def entry(name, nbytes, us, timing):
            e = {'bytes': nbytes, 'us': us, 'frac': nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 'timing': timing}
            t = tk.get(name, {}).get('traffic_bytes_per_launch')
            if t:                                                        # counter traffic vs algorithmic bytes (> 1: wasted re-reads)
                e['traffic'] = t
                e['traffic_over_algorithmic'] = t / nbytes
            return e
        per_launch = {'bt_eliminate_tw_kernel_L0': entry('bt_eliminate_tw_kernel_L0', elim_bytes, elim0_s * 1e6,
                                                         'burst of 40 launches between one event pair; first LM iteration and fallback solves only')}
        if fused_us:
            per_launch['trial_elim_kernel'] = entry('trial_elim_kernel', alg_bytes, fused_us,
                                                    'burst of 40 launches between one event pair (islam_pvgo_trial_elim_burst)')
            per_launch['trial_elim_kernel']['segment_len_segments_workgroups'] = fused_info
        if sweep_us:
            per_launch['bt_downsweep_kernel'] = entry('bt_downsweep_kernel', sweep_bytes, sweep_us, 'event pair around the single launch (adds ~2 us)')
        iter_bytes = 5000 * N                                           # ~5.0 KB per node per LM iteration
        roofline = {'bound': 'hbm', 'kernel': dom_name, 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                    'traffic_source': (None if traffic is None else '%s (rocprofv3 PMC passes,' % tfile.get('file') + ' separate FETCH_SIZE / WRITE_SIZE '
                                       'runs of this command; valid for pvgo.hip sha256[:16] = %s)' % tfile.get('pvgo_hip_sha16')) if not (tfile or {}).get('stale')
                                      else tfile['note'],
                    'algorithmic_bytes_per_launch': alg_bytes, 'avg_launch_us': dom_s * 1e6,
                    'solve_launch_us': kern, 'solve_burst_us': solve_us, 'levels_n_m_P': levels,
                    'iteration': {'bytes': iter_bytes, 'us': us_iter, 'frac': iter_bytes / (us_iter * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                  'what': 'whole LM iteration (solve + trial + control + next linearisation), SURVEY 8(d) traffic model'},
                    'per_launch': per_launch,
                    'latency_model': latency_model(ops, Hd, Ho, rhs, levels, device, us_iter),
                    'note': 'latency / issue-bound: the up-sweep is a chain of dependent 9x9 block pivots (one wavefront per half segment), '
                            'the SE(3) linearisation in front of it runs on 25 lanes per CU'}
        value = trials / elapsed
        out = {
            'metric': 'pvgo_lm_iters_per_sec', 'value': value, 'unit': 'LM iters/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[3]: synthetic %d-frame KITTI-shape chain graph (N=%d nodes, E=%d links), '
                                   'full run_pvgo LM loop per step' % (N - 1, N, N - 1),
                       'loss_weight': list(LOSS_WEIGHT), 'radius': 1e4, 'parallelism': 'graph sharded over %d GPU(s)' % world},
            'lm_iters_per_step': trials / args.steps, 'optimizer_steps_per_step': steps_lm / args.steps,
            'us_per_lm_iter': elapsed / trials * 1e6,
            'timed_passes': {'passes': passes, 'pass_ms': {'median': elapsed * 1e3, 'min': min(pass_s) * 1e3, 'max': max(pass_s) * 1e3},
                             'what': 'the K = --steps steps are timed `passes` times back to back, each pass between barrier + synchronize; '
                                     'value / ms_per_step / us_per_lm_iter are those of the MEDIAN pass'},
            'rccl_ranks_seen': rccl_ranks_seen,
            'roofline': roofline,
        }
        if world == 1 and not force_sharded:
            for key, fn in (('reject_heavy', lambda: reject_heavy_rate(ops, prob, device)),
                            ('trust_region_moving', lambda: trust_region_moving_rate(ops, prob, prm, device))):
                try:
                    out[key] = fn()
                except Exception as e:
                    out[key] = {'error': repr(e)[:300]}
        if large is not None:
            out['large_graph'] = large
        if replicas is not None:
            out['independent_graphs'] = replicas
        if world > 1 or force_sharded:
            out['sharded'] = sharded_info
        if not args.no_frontend and world == 1:
            try:
                out['stereo_vio'] = vio_frames_per_sec(device)
            except Exception as e:           # the headline metric must still be reported
                out['stereo_vio'] = {'error': repr(e)[:300]}
            if not args.no_cpu_baseline and isinstance(out['stereo_vio'], dict) and 'error' not in out['stereo_vio']:
                try:
                    out['stereo_vio']['cpu_baseline'] = vio_cpu_baseline()
                    out['stereo_vio']['speedup_vs_cpu'] = out['stereo_vio']['value'] / out['stereo_vio']['cpu_baseline']['value']
                except Exception as e:
                    out['stereo_vio']['cpu_baseline'] = {'error': repr(e)[:300]}
            try:
                out['front_end_per_launch'] = front_end_kernel_rooflines(device)
            except Exception as e:
                out['front_end_per_launch'] = {'error': repr(e)[:300]}
        if not args.no_cpu_baseline and world == 1:          # reported baseline: rank 0 at N=1 only
            host = {k: v.cpu().numpy() for k, v in (('init_nodes', prob['init_nodes']), ('init_vels', prob['init_vels']),
                                                     ('vo_motions', prob['vo']), ('imu_drots', prob['drots']),
                                                     ('imu_dtrans', prob['dtrans']), ('imu_dvels', prob['dvels']),
                                                     ('dts', prob['dts']))}
            host['links'] = tr['links']
            out['cpu_baseline'] = cpu_baseline(host)
            out['speedup_vs_cpu_port'] = value / out['cpu_baseline']['value']
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        # RCCL's banner goes through C stdio (block-buffered on a pipe): flush it first so the JSON line is the LAST line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
