// What pvgo.hip (the single-GPU LM loop and the stage kernels) shares with pvgo_dist.hip (the sharded LM loop): the run-ahead
// gate, the LM control rules as they run on the device, and the stage entry points WITH a gate.  Internal to libislam_hip.so --
// the public islam_pvgo_* stage functions of include/islam_hip.h call the same code ungated.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace islam {

// Run-ahead gate.  The host enqueues LM iteration t+1 before it knows the outcome of trial t, assuming "accepted,
// continue".  Every kernel of an iteration carries the epoch it was enqueued under; the deciding lane bumps the device
// epoch (state[14]) on any other outcome, which turns the already-queued kernels into no-ops.
struct Gate { const double* ctl; double epoch; };
__device__ __forceinline__ bool gate_closed(const Gate& g) { return g.ctl != nullptr && g.ctl[14] != g.epoch; }

// ------------------------------------------------------------------------------------------
// state: [0] loss [1] last [2] damping [3] radius [4] down [5] quality [6] trial loss [7] qden
//        [8] reject_count [9] accepted [10] error [11] has_loss [15] branch TrustRegion.update took last (speculated_damping)
//        [16 .. 16 + reject_count] the dampings applied to the diagonal of the CURRENT linearisation, in order (PyPose's
//        `A.diagonal().add_(A.diagonal() * damping)` is cumulative over the retries of one optimizer.step(): entry 0 is the damping of
//        the step's first solve, every reject appends one) -- what a solve that keeps the stored diagonal untouched (LevelSrc::hist,
//        the fused trial + elimination kernel) applies instead of damping in place
constexpr int STATE_DOUBLES = 40;
constexpr int STATE_HIST = 16;
// report (host-visible copy written after every trial): same slots as seen by the step that just ran, [15] = sequence number
//        [12] optimizer steps [13] plateau patience count [14] run-ahead epoch (Gate)
// report[12] = verdict: 0 accepted & continue, 1 rejected (retry with more damping), 2 accepted & stop, 3 solver failed &
// continue (same iterate, new linearisation), 4 solver failed & stop, 5 accepted & continue but the damping the run-ahead solve
// speculated on is not the one TrustRegion.update produced (the solve must be redone);  report[13] = optimizer steps so far
struct TRParams { double high, low, up, down, factor, rmin, rmax; int reject; int max_steps, patience; double decreasing; };

// StopOnPlateau.step(loss) after a finished optimizer.step() (pvgo.py:172,177-180): returns 1 when the loop must stop
__device__ __forceinline__ int scheduler_step(double* __restrict__ st, const TRParams& tr, double last, double loss, double rejects) {
    int stop = 0;
    st[12] += 1.0;
    if (st[12] >= (double)tr.max_steps) stop = 1;
    if ((last - loss) < tr.decreasing) st[13] += 1.0; else st[13] = 0.0;
    if (st[13] >= (double)tr.patience) stop = 1;
    if (rejects >= (double)tr.reject) stop = 1;
    return stop;
}

// The damping the NEXT solve will use if this trial is accepted and TrustRegion.update takes the same branch as it did for the
// previous trial (state[15]: 0 radius x up, 1 radius kept, 2 radius x down; an LM run stays in one regime for many trials) --
// what the fused trial + elimination kernel speculates on: it needs the damping before the grid-wide sums of the trial exist.
// The same floating-point operations as lm_control, so a correct guess is bit-identical to the real update.
__device__ __forceinline__ double speculated_damping(const double* __restrict__ st, const TRParams& tr) {
    const int cls = (int)st[15];
    double radius = 1.0 / st[2];
    if (cls == 0) radius = tr.up * radius;
    else if (cls == 2) radius = radius * st[4];
    radius = fmax(tr.rmin, fmin(radius, tr.rmax));
    return 1.0 / radius;
}

// pp.optim.LM accept/reject + ppost.TrustRegion.update on the summed partials (one lane)
// failed: a solver error was flagged for the solve whose trial this is.  d_spec >= 0: the solve of the next step is already running
// with this damping; an accepted trial whose TrustRegion.update yields another damping closes the gate (verdict 5).
// Returns the verdict (= report[12]); report may be nullptr (small_lm_kernel keeps the whole loop on the device).
// plain_report: `report` is device memory that a LATER launch reads (the sharded loop's down-sweep forwards it to the host): plain
// stores, no completion wait -- the system-scope write-through + wait costs the one-workgroup decision kernel ~1.5 us.
__device__ inline int lm_control(double s, double q, double* __restrict__ st, bool failed, const TRParams& tr,
                                 double* __restrict__ report, double seq, double d_spec = -1.0, bool plain_report = false) {
    double rep[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) rep[i] = 0.0;
    if (failed) {                        // solver failed: PyPose prints and breaks the step, nothing changes
        rep[0] = st[0]; rep[1] = st[1]; rep[2] = st[2]; rep[8] = st[8]; rep[10] = 1.0;
        rep[12] = scheduler_step(st, tr, st[1], st[0], st[8]) ? 4.0 : 3.0;
        st[8] = 0.0;
        st[STATE_HIST] = st[2];          // the next optimizer.step() rebuilds A and damps it once with the unchanged damping
        st[14] += 1.0;
    } else {
        const double last = st[1];
        const double quality = (last - s) / (-q);
        double radius = 1.0 / st[2], down = st[4];
        if (quality > tr.high) { radius = tr.up * radius; down = tr.down; st[15] = 0.0; }
        else if (quality > tr.low) { down = tr.down; st[15] = 1.0; }
        else { radius = radius * down; down = down * tr.factor; st[15] = 2.0; }
        down = fmax(tr.rmin, fmin(down, tr.rmax));
        radius = fmax(tr.rmin, fmin(radius, tr.rmax));
        st[3] = radius; st[4] = down; st[2] = 1.0 / radius; st[5] = quality; st[6] = s; st[7] = -q;
        rep[1] = last; rep[2] = st[2]; rep[3] = radius; rep[4] = down; rep[5] = quality; rep[6] = s; rep[7] = -q;
        if (last < s && st[8] < (double)tr.reject) {       // reject: the host keeps the old iterate, loss = last
            st[0] = last;
            st[8] += 1.0;
            st[STATE_HIST + min((int)st[8], STATE_DOUBLES - STATE_HIST - 1)] = st[2];      // the retry damps the already damped diagonal once more
            rep[0] = last; rep[8] = st[8]; rep[9] = 0.0;
            rep[12] = 1.0;
            st[14] += 1.0;
        } else {                                           // step kept (also when the reject limit is exhausted)
            rep[0] = s; rep[8] = st[8]; rep[9] = 1.0;
            const int stop = scheduler_step(st, tr, last, s, st[8]);
            const bool respec = !stop && d_spec >= 0.0 && st[2] != d_spec;
            rep[12] = stop ? 2.0 : (respec ? 5.0 : 0.0);
            if (stop || respec) st[14] += 1.0;
            st[0] = s;
            st[1] = s;                                     // next optimizer.step(): self.last = self.loss
            st[8] = 0.0;
            st[STATE_HIST] = st[2];                        // a new linearisation: damped once, with the new damping
        }
    }
    rep[13] = st[12];
    if (report && plain_report) {
#pragma unroll
        for (int i = 0; i < 15; ++i) report[i] = rep[i];
        report[15] = seq;
    } else if (report) {
#pragma unroll
        for (int i = 0; i < 15; ++i) __hip_atomic_store(&report[i], rep[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // payload written through before the sequence number
        __hip_atomic_store(&report[15], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return (int)rep[12];
}

// ---- gated stage entry points (defined in pvgo.hip) --------------------------------------------------------------------------
// The reprojection factor on a rank's LOCAL stretch of the chain: `reproj` describes the whole graph, link0 = global index of
// the local link 0 (keypoints / targets are offset by it; pvgo.py:57's frozen first motion applies to GLOBAL link 0 only).
// red: ISLAM_REPROJ_REC doubles per link (islam_pvgo_reproj_reduce), at nodes (dx == nullptr) or at Exp(dx) * nodes
int reproj_reduce_gated(const double* nodes, const double* dx, int M, const islam_pvgo_reproj* reproj, int link0, double* red,
                        Gate gate, hipStream_t s);
// linearisation + normal equations of a chain of N nodes in one launch (linbuild_kernel): lin (42 x (N-1), component-major),
// loss_part (one partial sum per 63-node block), Hd / Ho (N x 81), rhs (N x 9); reproj != nullptr: red = the reduction at nodes
int linbuild_gated(const double* nodes, const double* vels, const double* poses, const double* drots, const double* dtrans,
                   const double* dvels, const double* dts, int N, const islam_pvgo_params* prm, double* lin, double* loss_part,
                   double* Hd, double* Ho, double* rhs, const double* red, const islam_pvgo_reproj* reproj, int link0, Gate gate,
                   hipStream_t s);
// islam_pvgo_shard_upsweep; state != nullptr: the damping is read from state[2] on the device
int shard_upsweep_gated(double* Hd, const double* Ho, const double* rhs, double damping, const double* state, int N,
                        const int seg_len[2], int world, int rank, int node0, void* workspace, size_t workspace_bytes,
                        double* exchange, bool zero_exchange, int* flags, Gate gate, hipStream_t s);
// islam_pvgo_shard_downsweep
int shard_downsweep_gated(const double* exchange, int N, const int seg_len[2], int world, int rank, int node0, void* workspace,
                          size_t workspace_bytes, double* dx, int* flags, Gate gate, hipStream_t s);
// islam_pvgo_trial on M links whose linearisation records are lin[c * lin_stride + k]; reproj != nullptr: red_lin / red_trial =
// the reductions at the linearisation point and at the trial point
int trial_gated(const double* nodes, const double* vels, const double* dx, const double* poses, const double* drots,
                const double* dtrans, const double* dvels, const double* dts, const double* lin, int lin_stride, int M,
                double* nodes_t, double* vels_t, double* part, const double* red_lin, const double* red_trial,
                const islam_pvgo_reproj* reproj, int link0, Gate gate, hipStream_t s);


// ---- the sharded loop on the fused trial + elimination kernel (defined in pvgo.hip; the collective comes from pvgo_dist.hip)
// fn: recv = sum over the ranks of send (count doubles), enqueued on s
struct ShardSum { int (*fn)(void* self, const double* send, double* recv, size_t count, hipStream_t s); void* self; };
size_t shard_fused_scratch_doubles(int N, int world);
// *taken = 0: the plan is not one the fused kernel covers (or ISLAM_SHARD_FUSED=0) and nothing was enqueued -- the caller runs the
// launch-per-stage loop.  Otherwise the LM loop has run (the stream is NOT drained): *out_nodes / *out_vels = the arrays that hold
// the accepted iterate (full-size, global rows; only rows [*own0, *own1) are this rank's to hand on).
int run_chain_sharded_fused(const ShardSum& red, int world, int rank, double* nodes, double* vels, const double* poses, const double* drots,
                            const double* dtrans, const double* dvels, const double* dts, int N, const islam_pvgo_params* prm,
                            void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes, islam_pvgo_result* res,
                            long long* exchanged_bytes, hipStream_t s, int* taken, const double** out_nodes, const double** out_vels,
                            int* own0, int* own1);

}  // namespace islam
