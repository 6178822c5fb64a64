// pvgo_dist.hip -- the LM loop of ONE chain graph sharded over the GPUs of a node, in the library, on RCCL.
//
// No reference counterpart (the reference is single-GPU).  One process per GPU; every rank calls
// islam_pvgo_run_chain_sharded with the SAME full-size inputs (the problem is 0.9 MB) and works on its own stretch of the chain.
// Graphs the fused trial + elimination kernel covers (the plans islam_pvgo_run_chain fuses; no reprojection factor) run on
// pvgo.hip's run_chain_sharded_fused: trial_elim_kernel over the rank's segments -> levels 1 .. xl -> ONE all-reduce per trial
// [interface blocks of the next solve | sum r^2 | sum JD.(2R+JD) | failed pivots | the cut nodes' diagonal parts] -> decision ->
// down-sweep; no halo rows (see the comment there).  Everything else takes the launch-per-stage loop of this file:
// One LM trial = one gated chain of launches on the rank's stream:
//   linbuild_kernel            linearise + normal equations of the local links (only after an accepted step)
//   bt_eliminate_tw_kernel x   up-sweep of the local sub-tree, levels 0 .. xl                  (shard_upsweep_gated)
//   ncclAllReduce #1           the interface blocks, 351 doubles per segment of the exchange level (64.6 KB at N = 5001 / 8 ranks)
//   bt_downsweep_kernel        levels above xl redundantly + root + the local back-substitution, one launch (shard_downsweep_gated)
//   trial_kernel               trial step on the local links
//   msg_kernel                 [sum r^2 | sum JD.(2R+JD) | failed pivots | one 10-double halo per rank]
//   ncclAllReduce #2
//   decide_kernel              pp.optim.LM's accept / reject rule, TrustRegion.update and StopOnPlateau ON THE DEVICE, replicated:
//                              every rank sees the same all-reduced scalars, takes the same decision and writes a 128-byte verdict
//                              to pinned host memory.
// As in the single-GPU loop (pvgo.hip) the host runs one trial AHEAD: it enqueues trial t+1 under the assumption "accepted,
// continue" before it has seen the verdict of trial t; every kernel carries the epoch it was enqueued under and returns at once
// if the deciding lane bumped the device epoch.  The collectives of a cancelled chain still run (on unchanged buffers) -- every
// rank takes the same decisions, so the ranks enqueue the same sequence of collectives.  No stream synchronisation and no
// device->host copy inside the loop.  Once, at the end: ncclAllReduce of the assembled solution (N x 10 doubles).
// The same protocol as islam_amd/dist_pvgo.py (which drives the stage entry points from Python with host-side control and is
// what the gloo / virtual-rank tests exercise).  comm == NULL: world 1.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.h"
#include "pvgo_internal.h"

using namespace islam;

namespace {

constexpr int LIN_C = 42;          // components per link of the linearisation buffer (pvgo.hip)

#define ISLAM_NCCL_CHECK(expr)                                                                                    \
    do {                                                                                                          \
        ncclResult_t _r = (expr);                                                                                 \
        if (_r != ncclSuccess) return ::islam::fail(ISLAM_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
    } while (0)

// unweighted loss of the local links [0, n_own): rows of the residuals in the component-major linearisation buffer
__global__ __launch_bounds__(256) void own_loss_kernel(const double* __restrict__ lin, int M, int n_own, const double* __restrict__ rpred,
                                                       double* __restrict__ out) {
    __shared__ double red[256];
    const int rows[15] = {0, 1, 2, 3, 4, 5, 24, 25, 26, 36, 37, 38, 39, 40, 41};
    double s = 0.0;
    for (int k = threadIdx.x; k < n_own; k += 256) {
#pragma unroll
        for (int r = 0; r < 15; ++r) { const double v = lin[(size_t)rows[r] * M + k]; s = fma(v, v, s); }
        if (rpred) s += rpred[(size_t)k * ISLAM_REPROJ_REC + 27];                  // r^T r of the link's reprojection rows
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0];
}

// after all-reduce #2: the halo row of the trial iterate (the next rank's first node) and the LM decision -- identical on every
// rank, because the summed message is
__device__ __forceinline__ void decide(const double* __restrict__ msg, double* __restrict__ st, int* __restrict__ flags, const TRParams& tr,
                                       double* __restrict__ report, double seq, double* __restrict__ nt, double* __restrict__ vt,
                                       const double* __restrict__ halo, int halo_row, int t) {
    if (halo && t < 7) nt[(size_t)halo_row * 7 + t] = halo[t];
    if (halo && t >= 7 && t < 10) vt[(size_t)halo_row * 3 + (t - 7)] = halo[t];
    if (t == 0) {
        flags[0] = 0;
        lm_control(msg[0], msg[1], st, msg[2] > 0.0, tr, report, seq);   // a failed pivot on ANY rank fails the step everywhere
    }
}

// msg = [sum r^2 | sum JD.(2R+JD) | failed | halo of every rank (10 each)]: own part filled, the rest zero.  One wavefront.
// decide_now (world 1: nothing to sum over ranks): the decision is taken here.
__global__ __launch_bounds__(64) void msg_kernel(const double* __restrict__ part, int nblk, int* __restrict__ flags,
                                                 const double* __restrict__ nt_c, const double* __restrict__ vt_c, int first_local, int rank,
                                                 int world, double* __restrict__ msg, int decide_now, double* __restrict__ st, TRParams tr,
                                                 double* __restrict__ report, double seq, double* __restrict__ nt, double* __restrict__ vt,
                                                 Gate gate) {
    if (gate_closed(gate)) return;
    const int t = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = t; i < nblk; i += 64) { s += part[2 * i]; q += part[2 * i + 1]; }       // fixed pattern: deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); q += __shfl_down(q, o, 64); }
    if (decide_now) {
        if (t == 0) {
            msg[0] = s; msg[1] = q; msg[2] = flags[0] ? 1.0 : 0.0;
            decide(msg, st, flags, tr, report, seq, nt, vt, nullptr, 0, 0);
        }
        return;
    }
    for (int i = 3 + t; i < 3 + 10 * world; i += 64) msg[i] = 0.0;
    if (t == 0) { msg[0] = s; msg[1] = q; msg[2] = flags[0] ? 1.0 : 0.0; flags[0] = 0; }
    __syncthreads();
    if (t >= 3 && t < 10) msg[3 + 10 * rank + (t - 3)] = nt_c[(size_t)first_local * 7 + (t - 3)];
    if (t >= 10 && t < 13) msg[3 + 10 * rank + 7 + (t - 10)] = vt_c[(size_t)first_local * 3 + (t - 10)];
}

__global__ void decide_kernel(const double* __restrict__ msg, double* __restrict__ st, int* __restrict__ flags, TRParams tr,
                              double* __restrict__ report, double seq, double* __restrict__ nt, double* __restrict__ vt,
                              const double* __restrict__ halo, int halo_row, Gate gate) {
    if (gate_closed(gate)) return;
    decide(msg, st, flags, tr, report, seq, nt, vt, halo, halo_row, threadIdx.x);
}

// state <- [damping = 1 / radius (TrustRegion), radius, down, run-ahead epoch 1], everything else zero
__global__ void state_init_kernel(double* __restrict__ st, double radius, double down) {
    const int t = threadIdx.x;
    if (t < STATE_DOUBLES) st[t] = (t == 2 || t == STATE_HIST) ? 1.0 / radius : t == 3 ? radius : t == 4 ? down : t == 14 ? 1.0 : 0.0;
}

// error path: an epoch no enqueued chain carries turns everything still queued into no-ops
__global__ void close_gate_kernel(double* __restrict__ st, double epoch) {
    if (threadIdx.x == 0) st[14] = epoch;
}

// the loss of the very first optimizer.step() (summed over the ranks in msg[0])
__global__ void begin_kernel(const double* __restrict__ msg, double* __restrict__ st, int* __restrict__ flags) {
    if (threadIdx.x != 0) return;
    st[0] = msg[0]; st[1] = msg[0]; st[8] = 0.0; st[11] = 1.0; st[12] = 0.0; st[13] = 0.0;
    flags[0] = 0;
}

__global__ __launch_bounds__(256) void scatter_full_kernel(const double* __restrict__ nodes, const double* __restrict__ vels, int r0, int r1,
                                                            int a, double* __restrict__ full) {
    const int i = blockIdx.x * 256 + threadIdx.x;               // local rows [r0, r1) -> full rows a + r
    const int n = (r1 - r0) * 10;
    if (i >= n) return;
    const int r = r0 + i / 10, c = i % 10;
    full[(size_t)(a + r) * 10 + c] = c < 7 ? nodes[(size_t)r * 7 + c] : vels[(size_t)r * 3 + (c - 7)];
}

__global__ __launch_bounds__(256) void unpack_full_kernel(const double* __restrict__ full, int N, double* __restrict__ nodes, double* __restrict__ vels) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * 10) return;
    const int r = i / 10, c = i % 10;
    if (c < 7) nodes[(size_t)r * 7 + c] = full[i]; else vels[(size_t)r * 3 + (c - 7)] = full[i];
}

// the collective: RCCL in production; tests inject a callback (virtual ranks in threads on one GPU)
struct Reducer {
    ncclComm_t comm = nullptr;
    islam_allreduce_fn fn = nullptr;
    void* user = nullptr;
    // recv = sum over the ranks of send (send is left untouched)
    int sum_to(const double* send, double* recv, size_t count, hipStream_t s) const {
        if (fn) {
            ISLAM_HIP_CHECK(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, s));
            return sum(recv, count, s);
        }
        ISLAM_NCCL_CHECK(ncclAllReduce(send, recv, count, ncclDouble, ncclSum, comm, s));
        return ISLAM_OK;
    }
    int sum(double* buf, size_t count, hipStream_t s) const {
        if (fn) {
            if (fn(user, buf, count, (void*)s) != 0) return fail(ISLAM_EHIP, "islam_pvgo_run_chain_sharded: all-reduce callback failed");
            return ISLAM_OK;
        }
        ISLAM_NCCL_CHECK(ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, s));
        return ISLAM_OK;
    }
};

struct Shard { int xl, Pxl, seg0, nseg, node0, node1, n_own, first, has_left, has_right; };

int shard_of(int N, const int seg_len[2], int world, int rank, Shard& sh) {
    int out[2 + 2 * ISLAM_PVGO_MAX_LEVELS], plan[3 * ISLAM_PVGO_MAX_LEVELS + 1];
    const int rc = islam_pvgo_shard_ranges(N, seg_len, world, rank, out);
    if (rc != ISLAM_OK) return rc;
    islam_pvgo_plan(N, seg_len, plan);
    const int m = plan[1], stride = m + 1;
    sh.xl = out[0]; sh.Pxl = out[1]; sh.seg0 = out[2]; sh.nseg = out[3];
    sh.first = sh.seg0 * stride;
    const int sR = (sh.seg0 + sh.nseg - 1) * stride + m;
    sh.has_left = sh.seg0 > 0;
    sh.has_right = sR < N;
    sh.node0 = sh.has_left ? sh.first - 1 : 0;
    sh.node1 = sh.has_right ? std::min(sR + 1, N - 1) : N - 1;
    sh.n_own = (sh.has_right ? sR : N - 1) - sh.node0;
    return ISLAM_OK;
}

inline size_t a256(size_t n) { return align_up(n * sizeof(double)) / sizeof(double); }

}  // namespace

extern "C" {

int islam_dist_unique_id(void* out128) {
    ncclUniqueId id;
    ISLAM_NCCL_CHECK(ncclGetUniqueId(&id));
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out128, &id, sizeof id);
    return ISLAM_OK;
}

int islam_dist_comm_init(const void* id128, int world, int rank, void** comm) {
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    ISLAM_NCCL_CHECK(ncclCommInitRank(&c, world, id, rank));
    *comm = c;
    return ISLAM_OK;
}

int islam_dist_comm_destroy(void* comm) {
    if (comm) ISLAM_NCCL_CHECK(ncclCommDestroy((ncclComm_t)comm));
    return ISLAM_OK;
}

int islam_dist_comm_info(void* comm, int* ranks, int* rank, int* device) {
    if (!comm) {                       // world 1 runs without a communicator
        if (ranks) *ranks = 1;
        if (rank) *rank = 0;
        if (device) ISLAM_HIP_CHECK(hipGetDevice(device));
        return ISLAM_OK;
    }
    int v = 0;
    if (ranks) { ISLAM_NCCL_CHECK(ncclCommCount((ncclComm_t)comm, &v)); *ranks = v; }
    if (rank) { ISLAM_NCCL_CHECK(ncclCommUserRank((ncclComm_t)comm, &v)); *rank = v; }
    if (device) { ISLAM_NCCL_CHECK(ncclCommCuDevice((ncclComm_t)comm, &v)); *device = v; }
    return ISLAM_OK;
}

size_t islam_pvgo_sharded_scratch_bytes(int N, int world) {
    const size_t n = (size_t)N + 2;
    size_t d = a256(LIN_C * n) + a256(n / 8 + 4) + 2 * a256(81 * n) + 2 * a256(9 * n) + 2 * a256(7 * n) + 2 * a256(3 * n) +
               a256(2 * (n / 64 + 2)) + a256(3 + 10 * (size_t)world) + a256(10 * n) + 2 * a256(351 * (n / 5 + 2)) + a256(64) + a256(STATE_DOUBLES) + 2 * a256(ISLAM_REPROJ_REC * n);
    return d * sizeof(double) + 512;
}

static int run_sharded(const Reducer& red, int world, int rank, double* nodes, double* vels, const double* poses, const double* drots,
                       const double* dtrans, const double* dvels, const double* dts, int N, const islam_pvgo_params* prm,
                       const islam_pvgo_reproj* reproj, void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                       islam_pvgo_result* res, long long* exchanged_bytes, void* stream) {
    if (!prm || !res || N < 2 || world < 1 || rank < 0 || rank >= world) return fail(ISLAM_EARG, "islam_pvgo_run_chain_sharded: bad argument");
    if (world > 1 && !red.comm && !red.fn) return fail(ISLAM_EARG, "islam_pvgo_run_chain_sharded: world=%d needs a communicator", world);
    if (scratch_bytes < islam_pvgo_sharded_scratch_bytes(N, world)) return fail(ISLAM_EARG, "islam_pvgo_run_chain_sharded: scratch too small");
    Shard sh;
    int rc = shard_of(N, prm->seg_len, world, rank, sh);
    if (rc != ISLAM_OK) return rc;
    hipStream_t s = as_stream(stream);
    const int a = sh.node0, b = sh.node1, nloc = b - a + 1, Mloc = nloc - 1, n_own = sh.n_own;
    const int first_local = sh.first - a, nblk = (n_own + 63) / 64, nmsg = 3 + 10 * world;
    // scratch carve
    double* p = (double*)align_up((size_t)scratch);
    auto take = [&](size_t n) { double* r = p; p += a256(n); return r; };
    const size_t nn = (size_t)N + 2;
    double* lin = take(LIN_C * nn);
    double* loss_part = take(nn / 8 + 4);
    double* Hd = take(81 * nn); double* Ho = take(81 * nn); double* rhs = take(9 * nn); double* dx = take(9 * nn);
    double* nl = take(7 * nn); double* nt = take(7 * nn); double* vl = take(3 * nn); double* vt = take(3 * nn);
    double* part = take(2 * (nn / 64 + 2));
    double* msg = take(nmsg);
    double* full = take(10 * nn);
    double* ex_own = take(351 * (nn / 5 + 2));        // own rows of the exchange level, everything else stays zero (zeroed once)
    double* ex = world > 1 ? take(351 * (nn / 5 + 2)) : ex_own;      // the sum over the ranks
    int* flags = (int*)take(64);
    double* state = take(STATE_DOUBLES);
    double* rp_lin = take(ISLAM_REPROJ_REC * nn);      // reprojection factor: per-link reductions at the linearisation point ...
    double* rp_tri = take(ISLAM_REPROJ_REC * nn);      // ... and at the trial point
    // verdicts in pinned, device-visible host memory (two slots, alternating with the trial number); the host polls the
    // sequence number -- no stream synchronisation, no copy
    static thread_local double* host_state = nullptr;
    if (!host_state) ISLAM_HIP_CHECK(hipHostMalloc((void**)&host_state, 32 * sizeof(double), hipHostMallocMapped | hipHostMallocPortable));   // portable: one buffer per thread serves calls on any device
    double* report = nullptr;
    ISLAM_HIP_CHECK(hipHostGetDevicePointer((void**)&report, host_state, 0));
    volatile double* hs_all = host_state;
    hs_all[15] = 0.0;
    hs_all[31] = 0.0;
    // ---- the loop on the fused trial + elimination kernel, one collective per trial (pvgo.hip: run_chain_sharded_fused); plans it
    // does not cover and the reprojection factor take the launch-per-stage loop below
    if (!reproj) {
        ShardSum sum{[](void* self, const double* send, double* recv, size_t count, hipStream_t st) -> int {
                         return static_cast<const Reducer*>(self)->sum_to(send, recv, count, st);
                     },
                     const_cast<Reducer*>(&red)};
        int taken = 0, own0 = 0, own1 = 0;
        const double *fin_n = nullptr, *fin_v = nullptr;
        rc = run_chain_sharded_fused(sum, world, rank, nodes, vels, poses, drots, dtrans, dvels, dts, N, prm, workspace, workspace_bytes, scratch,
                                     scratch_bytes, res, exchanged_bytes, s, &taken, &fin_n, &fin_v, &own0, &own1);
        if (rc != ISLAM_OK) return rc;
        if (taken) {
            if (world == 1) {
                if (fin_n != nodes) {
                    ISLAM_HIP_CHECK(hipMemcpyAsync(nodes, fin_n, sizeof(double) * 7 * (size_t)N, hipMemcpyDeviceToDevice, s));
                    ISLAM_HIP_CHECK(hipMemcpyAsync(vels, fin_v, sizeof(double) * 3 * (size_t)N, hipMemcpyDeviceToDevice, s));
                }
            } else {
                ISLAM_HIP_CHECK(hipMemsetAsync(full, 0, sizeof(double) * 10 * (size_t)N, s));
                hipLaunchKernelGGL(scatter_full_kernel, dim3(((own1 - own0) * 10 + 255) / 256), dim3(256), 0, s, fin_n, fin_v, own0, own1, 0, full);
                if ((rc = red.sum(full, 10 * (size_t)N, s)) != ISLAM_OK) return rc;
                hipLaunchKernelGGL(unpack_full_kernel, dim3((N * 10 + 255) / 256), dim3(256), 0, s, full, N, nodes, vels);
                ISLAM_LAUNCH_CHECK();
                ISLAM_HIP_CHECK(hipStreamSynchronize(s));       // (as the launch-per-stage loop below: the ranks leave together)
            }
            // (one rank: stream-ordered like islam_pvgo_run_chain -- the result block comes from the pinned verdicts, nothing to wait for)
            return ISLAM_OK;
        }
    }
    hipLaunchKernelGGL(state_init_kernel, dim3(1), dim3(64), 0, s, state, prm->radius, prm->down);     // (no host->device copy: see pvgo.hip)
    ISLAM_HIP_CHECK(hipMemsetAsync(flags, 0, 64 * sizeof(double), s));
    ISLAM_HIP_CHECK(hipMemsetAsync(workspace, 0, workspace_bytes, s));     // product rows of other ranks' segments read as zero
    ISLAM_HIP_CHECK(hipMemsetAsync(ex_own, 0, sizeof(double) * 351 * (size_t)sh.Pxl, s));
    ISLAM_HIP_CHECK(hipMemcpyAsync(nl, nodes + (size_t)a * 7, sizeof(double) * 7 * nloc, hipMemcpyDeviceToDevice, s));
    ISLAM_HIP_CHECK(hipMemcpyAsync(vl, vels + (size_t)a * 3, sizeof(double) * 3 * nloc, hipMemcpyDeviceToDevice, s));
    ISLAM_HIP_CHECK(hipMemcpyAsync(nt, nl, sizeof(double) * 7 * nloc, hipMemcpyDeviceToDevice, s));     // rows no trial writes (a last
    ISLAM_HIP_CHECK(hipMemcpyAsync(vt, vl, sizeof(double) * 3 * nloc, hipMemcpyDeviceToDevice, s));     // rank's unused tail) stay defined
    const double *lp = poses + (size_t)a * 7, *lr = drots + (size_t)a * 4, *ltr = dtrans + (size_t)a * 3, *lv = dvels + (size_t)a * 3,
                 *ldt = dts + a;
    const TRParams tr{prm->high, prm->low, prm->up, prm->down, prm->factor, prm->rmin, prm->rmax, prm->reject,
                      prm->max_steps, prm->patience, prm->decreasing};
    const bool halo = sh.has_right && rank + 1 < world && b > a + n_own;
    const long long iter_bytes = (world > 1) ? 8LL * (351LL * sh.Pxl + nmsg) : 0;
    long long xbytes = 0;

    struct IterCfg { double *cur_n, *cur_v, *tri_n, *tri_v; };
    // one pass of PyPose's inner `while self.last <= self.loss`; relin: the iterate changed, linearise first
    auto enqueue_iter = [&](const IterCfg& c, double seq, double epoch, bool relin) -> int {
        const Gate gate{state, epoch};
        int r;
        if (relin) {
            if (reproj && (r = reproj_reduce_gated(c.cur_n, nullptr, Mloc, reproj, a, rp_lin, gate, s)) != ISLAM_OK) return r;
            if ((r = linbuild_gated(c.cur_n, c.cur_v, lp, lr, ltr, lv, ldt, nloc, prm, lin, loss_part, Hd, Ho, rhs, rp_lin, reproj, a, gate, s)) != ISLAM_OK) return r;
        }
        if ((r = shard_upsweep_gated(Hd, Ho, rhs, 0.0, state, N, prm->seg_len, world, rank, a, workspace, workspace_bytes, ex_own, false, flags, gate, s)) != ISLAM_OK) return r;
        if (world > 1 && (r = red.sum_to(ex_own, ex, 351 * (size_t)sh.Pxl, s)) != ISLAM_OK) return r;
        if ((r = shard_downsweep_gated(ex, N, prm->seg_len, world, rank, a, workspace, workspace_bytes, dx, flags, gate, s)) != ISLAM_OK) return r;
        // (the reduction at the trial point covers the rank's own links: the step of the halo node is not known here)
        if (reproj && (r = reproj_reduce_gated(c.cur_n, dx, n_own, reproj, a, rp_tri, gate, s)) != ISLAM_OK) return r;
        if ((r = trial_gated(c.cur_n, c.cur_v, dx, lp, lr, ltr, lv, ldt, lin, Mloc, n_own, c.tri_n, c.tri_v, part, rp_lin, rp_tri, reproj, a, gate, s)) != ISLAM_OK) return r;
        double* rep = report + 16 * ((long long)seq & 1);
        hipLaunchKernelGGL(msg_kernel, dim3(1), dim3(64), 0, s, part, nblk, flags, c.tri_n, c.tri_v, first_local, rank, world, msg,
                           world == 1 ? 1 : 0, state, tr, rep, seq, c.tri_n, c.tri_v, gate);
        if (world > 1) {
            if ((r = red.sum(msg, nmsg, s)) != ISLAM_OK) return r;
            hipLaunchKernelGGL(decide_kernel, dim3(1), dim3(64), 0, s, msg, state, flags, tr, rep, seq, c.tri_n, c.tri_v,
                               halo ? msg + 3 + 10 * (rank + 1) : (const double*)nullptr, b - a, gate);
        }
        ISLAM_LAUNCH_CHECK();
        xbytes += iter_bytes;
        return ISLAM_OK;
    };

    int steps = 0, trials = 0, status = ISLAM_OK;
    double loss = 0.0, damping = 1.0 / prm->radius, epoch = 1.0;
    IterCfg A{nl, vl, nt, vt};            // the iteration whose verdict is awaited
    auto run = [&]() -> int {
    // first linearisation and the loss of the initial iterate
    const Gate open{nullptr, 0.0};
    if (reproj && (rc = reproj_reduce_gated(nl, nullptr, Mloc, reproj, a, rp_lin, open, s)) != ISLAM_OK) return rc;
    if ((rc = linbuild_gated(nl, vl, lp, lr, ltr, lv, ldt, nloc, prm, lin, loss_part, Hd, Ho, rhs, rp_lin, reproj, a, open, s)) != ISLAM_OK) return rc;
    hipLaunchKernelGGL(own_loss_kernel, dim3(1), dim3(256), 0, s, lin, Mloc, n_own, reproj ? rp_lin : (const double*)nullptr, msg);
    if (world > 1 && (rc = red.sum(msg, 1, s)) != ISLAM_OK) return rc;
    hipLaunchKernelGGL(begin_kernel, dim3(1), dim3(64), 0, s, msg, state, flags);

    if ((rc = enqueue_iter(A, 1.0, epoch, false)) != ISLAM_OK) return rc;
    for (;;) {
        const double seq = (double)(trials + 1);
        // run ahead: the next iteration under the assumption "trial accepted, loop continues"
        const IterCfg B{A.tri_n, A.tri_v, A.cur_n, A.cur_v};
        if (steps + 1 < prm->max_steps && (rc = enqueue_iter(B, seq + 1.0, epoch, true)) != ISLAM_OK) return rc;
        volatile double* hs = hs_all + 16 * ((long long)seq & 1);
        {
            unsigned long spins = 0;
            while (hs[15] != seq) {
                if (++spins > 400000000ul) {
                    ISLAM_HIP_CHECK(hipStreamSynchronize(s));
                    if (hs[15] != seq) return fail(ISLAM_EHIP, "islam_pvgo_run_chain_sharded: no status from the device (trial %d)", trials + 1);
                }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        ++trials;
        const int verdict = (int)hs[12];
        damping = hs[2];
        loss = hs[0];
        steps = (int)hs[13];
        if (verdict == 0) { A = B; continue; }          // accepted, continue: B is the iteration now in flight
        epoch += 1.0;                                   // any other verdict bumped the device epoch: B's kernels are no-ops
        if (verdict == 1) {                             // rejected: same iterate, same (cumulatively damped) linearisation
            if ((rc = enqueue_iter(A, seq + 1.0, epoch, false)) != ISLAM_OK) return rc;
            continue;
        }
        if (verdict == 2) { A = B; break; }             // accepted, StopOnPlateau says stop
        status = ISLAM_ENOTPD;                          // "Linear solver failed. Breaking optimization step..."
        if (verdict == 4) break;
        if ((rc = enqueue_iter(A, seq + 1.0, epoch, true)) != ISLAM_OK) return rc;      // same iterate, new linearisation
    }
    // the full solution on every rank: own rows (the left outer separator belongs to the previous rank), summed
    ISLAM_HIP_CHECK(hipMemsetAsync(full, 0, sizeof(double) * 10 * (size_t)N, s));
    const int r0 = sh.has_left ? 1 : 0, r1 = n_own + 1;
    hipLaunchKernelGGL(scatter_full_kernel, dim3(((r1 - r0) * 10 + 255) / 256), dim3(256), 0, s, A.cur_n, A.cur_v, r0, r1, a, full);
    if (world > 1 && (rc = red.sum(full, 10 * (size_t)N, s)) != ISLAM_OK) return rc;
    hipLaunchKernelGGL(unpack_full_kernel, dim3((N * 10 + 255) / 256), dim3(256), 0, s, full, N, nodes, vels);
    ISLAM_LAUNCH_CHECK();
    ISLAM_HIP_CHECK(hipStreamSynchronize(s));
    return ISLAM_OK;
    };
    rc = run();
    if (rc != ISLAM_OK) {
        // A failed enqueue / collective / status wait leaves gated kernels (and possibly collectives) of the run-ahead chain in
        // flight: they would still write the pinned verdict block and the scratch the NEXT call reuses.  Close the gate (any
        // epoch no enqueued kernel carries) and drain the stream before handing the error up; the message of the original
        // failure is kept.  After a collective error the peers may be blocked in their next collective: the caller must destroy
        // (abort) the communicator.
        hipLaunchKernelGGL(close_gate_kernel, dim3(1), dim3(64), 0, s, state, epoch + 2.0);
        (void)hipStreamSynchronize(s);
        (void)hipGetLastError();
        return rc;
    }
    res->steps = steps; res->trials = trials; res->status = status; res->loss = loss; res->damping = damping;
    if (exchanged_bytes) *exchanged_bytes = xbytes;
    return ISLAM_OK;
}

int islam_pvgo_run_chain_sharded(void* comm, int world, int rank, double* nodes, double* vels, const double* poses, const double* drots,
                                 const double* dtrans, const double* dvels, const double* dts, int N, const islam_pvgo_params* prm,
                                 const islam_pvgo_reproj* reproj, void* workspace, size_t workspace_bytes, void* scratch,
                                 size_t scratch_bytes, islam_pvgo_result* res, long long* exchanged_bytes, void* stream) {
    Reducer red;
    red.comm = (ncclComm_t)comm;
    return run_sharded(red, world, rank, nodes, vels, poses, drots, dtrans, dvels, dts, N, prm, reproj, workspace, workspace_bytes, scratch,
                       scratch_bytes, res, exchanged_bytes, stream);
}

int islam_pvgo_run_chain_sharded_cb(islam_allreduce_fn fn, void* user, int world, int rank, double* nodes, double* vels,
                                    const double* poses, const double* drots, const double* dtrans, const double* dvels, const double* dts,
                                    int N, const islam_pvgo_params* prm, const islam_pvgo_reproj* reproj, void* workspace,
                                    size_t workspace_bytes, void* scratch, size_t scratch_bytes, islam_pvgo_result* res,
                                    long long* exchanged_bytes, void* stream) {
    Reducer red;
    red.fn = fn;
    red.user = user;
    return run_sharded(red, world, rank, nodes, vels, poses, drots, dtrans, dvels, dts, N, prm, reproj, workspace, workspace_bytes, scratch,
                       scratch_bytes, res, exchanged_bytes, stream);
}

}  // extern "C"
