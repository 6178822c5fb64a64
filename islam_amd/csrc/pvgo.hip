// PVGO back-end on gfx950: SE(3) Levenberg-Marquardt over VO / IMU factors on a chain graph, fp64.
//
// Replaces (reference file:line under /root/reference):
//   pvgo.py:26-64    PoseVelGraph.forward        -> linearize_kernel / trial_kernel (residuals)
//   pvgo.py:168-180  pp.optim.LM + TrustRegion + StopOnPlateau loop -> islam_pvgo_run_chain
//   pvgo.py:67-78    vo_loss (+ PyPose backward)  -> vo_loss_{fwd,bwd}_kernel
//   pvgo.py:114-119  align_to                     -> align_kernel
//   PyPose (external, un-vendored): modjac/jacrev Jacobian, J^T W J, cholesky_ex/cholesky_solve,
//   LieTensor.add_ -- restated from its published algorithm (SURVEY.md section 8a box).
//
// Design: the reference materialises a dense (rows x 10N) Jacobian, a dense
// block_diag weight and a dense 10N x 10N normal matrix.  The graph is a chain, so the normal
// matrix is block-tridiagonal with 9x9 blocks [rho phi v] per node.  We
//   1. linearise per link (one lane per link, everything in registers),
//   2. build the 9x9 diagonal / coupling blocks per node (gather from the two adjacent links,
//      no atomics, deterministic),
//   3. factor with a partitioned ("spike") block LDL^T: the chain is cut into segments, each
//      segment's interior is eliminated onto its two separator nodes -- by two wavefronts that start
//      at the two ends and meet at the middle node (bt_eliminate_tw_kernel, "twisted"; the one-sided
//      bt_eliminate_kernel serves segment lengths the twisted path does not handle) -- one launch per level; the separators
//      form a ~6x smaller chain that is reduced the same way;
//      the root solve and the whole back-substitution run in ONE launch (bt_downsweep_kernel) with
//      per-segment ready words and write-through hand-off between levels.  Inside a wavefront one
//      lane owns one column of the augmented 9 x 28 matrix [S | U | F^T | g]; pivots are broadcast
//      with v_readlane, the 19x19 Schur update runs over all 64 lanes from an LDS copy,
//   4. apply the step on a trial copy, re-evaluate the loss and the trust-region ratio per link AND
//      linearise at the trial point for the next step in the same launch (trial_lin_kernel); one
//      extra workgroup waits for all partial sums, takes the LM / TrustRegion / StopOnPlateau decision
//      and writes a 128-byte verdict to pinned host memory,
//   5. the host enqueues one iteration ahead (every kernel is gated on an epoch the deciding lane
//      bumps on any verdict other than "accepted, continue") and only polls the verdicts.
// linearize_kernel / build_normal_kernel / trial_kernel / bt_top_kernel / bt_backsub_kernel are the
// stage-level versions behind the islam_pvgo_* entry points the sharded driver and the tests call.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "lie_dev.h"
#include "pvgo_internal.h"

using namespace islam;

namespace {

constexpr int LIN_C = 42;   // doubles per link record (component-major: lin[c*M + k])
constexpr int FAC = 252;    // 28 columns x 9 rows stored per eliminated node
constexpr int XS = 10;      // padded column stride of the LDS copies (16-byte aligned columns)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

#ifdef ISLAM_PROBE
__device__ long long islam_probe_buf[1024];
#define PROBE_AT(cond, slot) do { __builtin_amdgcn_sched_barrier(0); if (cond) islam_probe_buf[(slot)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PROBE_WALL(cond, slot) do { __builtin_amdgcn_sched_barrier(0); if (cond) islam_probe_buf[(slot)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PROBE_AT(cond, slot) do { } while (0)
#define PROBE_WALL(cond, slot) do { } while (0)
#endif

// XCD-aware work mapping.  Workgroup b is observed to run on XCD b % 8 (MI355X_MICROARCH.md, dispatch section; speed
// only, never correctness).  Every kernel of the LM loop splits its node / link / segment range into 8 CONTIGUOUS parts,
// part x handled by the workgroups with b % 8 == x, so data produced for a stretch of the chain stays in the L2 of the XCD
// that consumes it in the next launch (the levels of the block solver hand ~4 KB per segment to each other).
// Launch with xcd_grid(total) workgroups; returns the logical index or -1 for the padding workgroups.
__host__ __device__ __forceinline__ int xcd_grid(int total) { return 8 * ((total + 7) / 8); }
__device__ __forceinline__ int xcd_index(int b, int total) {
    const int Q = (total + 7) / 8;
    const int s = (b & 7) * Q + (b >> 3);
    return s < total ? s : -1;
}

__device__ __forceinline__ double bcast(double v, int src) {   // src must be wave-uniform
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

// Agent-coherent accesses (global_store / global_load with sc1: written through to / read from the level every XCD
// sees).  Data handed from one workgroup to another INSIDE a launch goes through these, so that publishing needs no
// release fence: an agent-scope release is a write-back walk of the XCD's whole L2 (microseconds when many waves do it).
__device__ __forceinline__ void st_coherent(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_coherent(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (the run-ahead gate -- struct Gate, gate_closed -- lives in pvgo_internal.h)

#include "pvgo_linearize.inl"   // link residuals, Jacobian blocks, normal equations, the sparse reprojection factor, linbuild_kernel
#include "pvgo_solver.inl"   // partitioned / twisted block-tridiagonal LDL^T
#include "pvgo_lm_kernels.inl"   // device side of the LM loop
#include "pvgo_general.inl"   // control kernels, retraction, arbitrary-topology assembly, vo_loss / imu_loss forward + backward, align_to
#include "pvgo_host.inl"   // host side
}  // namespace

extern "C" {

void islam_pvgo_default_params(islam_pvgo_params* p) {
    p->w[0] = p->w[1] = p->w[2] = p->w[3] = 1.0;
    p->radius = 1e4;
    p->vmin = 1e-4;
    p->vmax = 1e32;
    p->high = 0.5; p->low = 1e-3; p->up = 2.0; p->down = 0.5; p->factor = 0.5; p->rmin = 1e-6; p->rmax = 1e16;
    p->reject = 16;
    p->max_steps = 10;
    p->patience = 3;
    p->decreasing = 1e-3;
    p->seg_len[0] = p->seg_len[1] = 0;
}

size_t islam_pvgo_workspace_bytes(int N) {
    if (N < 1) return 0;
    return carve(nullptr, N).bytes + 256;
}

int islam_pvgo_linearize(const double* nodes, const double* vels, const double* poses, const double* drots,
                         const double* dtrans, const double* dvels, const double* dts, int N, double* lin,
                         double* loss_part, void* stream) {
    if (N < 2) return fail(ISLAM_EARG, "islam_pvgo_linearize: N=%d < 2", N);
    const int M = N - 1, nblk = (M + 63) / 64;
    hipLaunchKernelGGL(linearize_kernel, dim3(nblk), dim3(64), 0, as_stream(stream), nodes, vels, poses, drots, dtrans,
                       dvels, dts, M, lin, loss_part);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_pvgo_build_normal(const double* lin, const double* dts, int N, const double w[4], double vmin, double vmax,
                            double* Hd, double* Ho, double* rhs, void* stream) {
    if (N < 2) return fail(ISLAM_EARG, "islam_pvgo_build_normal: N=%d < 2", N);
    hipLaunchKernelGGL(build_normal_kernel, dim3((N + 63) / 64), dim3(64), 0, as_stream(stream), lin, dts, N, w[0], w[1],
                       w[2], w[3], vmin, vmax, Hd, Ho, rhs);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_pvgo_solve_chain(double* Hd, const double* Ho, const double* rhs, double damping, int N, const int seg_len[2],
                           void* workspace, size_t workspace_bytes, double* dx, void* stream) {
    if (N < 1) return fail(ISLAM_EARG, "islam_pvgo_solve_chain: N=%d < 1", N);
    if (workspace_bytes < islam_pvgo_workspace_bytes(N))
        return fail(ISLAM_EARG, "islam_pvgo_solve_chain: workspace %zu < %zu bytes", workspace_bytes,
                    islam_pvgo_workspace_bytes(N));
    Workspace w = carve((void*)align_up((size_t)workspace), N);
    hipStream_t s = as_stream(stream);
    ISLAM_HIP_CHECK(hipMemsetAsync(w.flags, 0, 4 * sizeof(double), s));
    ISLAM_HIP_CHECK(hipMemsetAsync(w.ready, 0, w.ready_bytes, s));
    int rc = enqueue_solve(w, Hd, Ho, rhs, nullptr, damping, N, seg_len, dx, s);
    if (rc != ISLAM_OK) return rc;
    int flag = 0;
    ISLAM_HIP_CHECK(hipMemcpyAsync(&flag, w.flags, sizeof(int), hipMemcpyDeviceToHost, s));
    ISLAM_HIP_CHECK(hipStreamSynchronize(s));
    if (flag) return fail(ISLAM_ENOTPD, "islam_pvgo_solve_chain: non-positive pivot (matrix not positive definite)");
    return ISLAM_OK;
}

// Stream-ordered variant of islam_pvgo_solve_chain: enqueue only, no read-back and no synchronisation (an iterative method
// calls it hundreds of times with the same matrix).  islam_pvgo_solve_status() reports, after a stream synchronisation,
// whether ANY solve enqueued on this workspace since the last status call met a non-positive pivot.
int islam_pvgo_solve_chain_enqueue(double* Hd, const double* Ho, const double* rhs, double damping, int N, const int seg_len[2],
                                   void* workspace, size_t workspace_bytes, double* dx, void* stream) {
    if (N < 1) return fail(ISLAM_EARG, "islam_pvgo_solve_chain_enqueue: N=%d < 1", N);
    if (workspace_bytes < islam_pvgo_workspace_bytes(N))
        return fail(ISLAM_EARG, "islam_pvgo_solve_chain_enqueue: workspace %zu < %zu bytes", workspace_bytes,
                    islam_pvgo_workspace_bytes(N));
    Workspace w = carve((void*)align_up((size_t)workspace), N);
    return enqueue_solve(w, Hd, Ho, rhs, nullptr, damping, N, seg_len, dx, as_stream(stream));
}

// 0: every solve since the last call was positive definite; ISLAM_ENOTPD otherwise.  Also (re)initialises the workspace's
// status and hand-off words: call it once BEFORE the first islam_pvgo_solve_chain_enqueue on a fresh workspace.
int islam_pvgo_solve_status(int N, void* workspace, size_t workspace_bytes, void* stream) {
    if (N < 1 || workspace_bytes < islam_pvgo_workspace_bytes(N)) return fail(ISLAM_EARG, "islam_pvgo_solve_status: bad workspace");
    Workspace w = carve((void*)align_up((size_t)workspace), N);
    hipStream_t s = as_stream(stream);
    int flag = 0;
    ISLAM_HIP_CHECK(hipMemcpyAsync(&flag, w.flags, sizeof(int), hipMemcpyDeviceToHost, s));
    ISLAM_HIP_CHECK(hipStreamSynchronize(s));
    ISLAM_HIP_CHECK(hipMemsetAsync(w.flags, 0, 4 * sizeof(double), s));
    ISLAM_HIP_CHECK(hipMemsetAsync(w.ready, 0, w.ready_bytes, s));
    if (flag) return fail(ISLAM_ENOTPD, "islam_pvgo_solve_status: non-positive pivot (matrix not positive definite)");
    return ISLAM_OK;
}

// Profiling variant of islam_pvgo_solve_chain: HIP events around every launch of one solve, on the stream the
// kernels run on.  ms[i] = duration of launch i (eliminate level 0..L-1, then back-substitution L-2..0);
// plan[3*l+0..2] = (nodes, segment length, segments) of level l.  Returns the number of launches in *nlaunch.
int islam_pvgo_solve_chain_timed(double* Hd, const double* Ho, const double* rhs, double damping, int N,
                                 const int seg_len[2], void* workspace, size_t workspace_bytes, double* dx, float* ms,
                                 int* plan_out, int* nlaunch, void* stream) {
    if (N < 1) return fail(ISLAM_EARG, "islam_pvgo_solve_chain_timed: N=%d < 1", N);
    if (workspace_bytes < islam_pvgo_workspace_bytes(N)) return fail(ISLAM_EARG, "islam_pvgo_solve_chain_timed: workspace too small");
    Workspace w = carve((void*)align_up((size_t)workspace), N);
    hipStream_t s = as_stream(stream);
    ISLAM_HIP_CHECK(hipMemsetAsync(w.ready, 0, w.ready_bytes, s));
    hipEvent_t evs[2 * MAXL + 2];
    for (auto& e : evs) ISLAM_HIP_CHECK(hipEventCreate(&e));
    ISLAM_HIP_CHECK(hipMemsetAsync(w.flags, 0, 4 * sizeof(double), s));
    int ne = 0;
    int rc = enqueue_solve(w, Hd, Ho, rhs, nullptr, damping, N, seg_len, dx, s, evs, &ne);
    if (rc != ISLAM_OK) return rc;
    ISLAM_HIP_CHECK(hipStreamSynchronize(s));
    for (int i = 0; i + 1 < ne; ++i) ISLAM_HIP_CHECK(hipEventElapsedTime(&ms[i], evs[i], evs[i + 1]));
    for (auto& e : evs) (void)hipEventDestroy(e);
    SolvePlan sp;
    const int nl = plan_levels(N, seg_len, sp, solve_twisted());
    for (int l = 0; l < MAXL; ++l) {
        plan_out[3 * l] = l < nl ? sp.lv[l].n : 0;
        plan_out[3 * l + 1] = l < nl ? sp.lv[l].m : 0;
        plan_out[3 * l + 2] = l < nl ? sp.lv[l].P : 0;
    }
    plan_out[3 * MAXL] = sp.top;
    *nlaunch = ne - 1;
    return ISLAM_OK;
}

// Measurement hook (bench.py's roofline leg): exactly the level-0 up-sweep launch of islam_pvgo_solve_chain -- same
// kernel, grid and arguments -- and nothing else.  Hd's diagonal is damped in place like in a solve.
int islam_pvgo_eliminate_level0(double* Hd, const double* Ho, const double* rhs, double damping, int N, const int seg_len[2],
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (N < 1) return fail(ISLAM_EARG, "islam_pvgo_eliminate_level0: N=%d < 1", N);
    if (workspace_bytes < islam_pvgo_workspace_bytes(N)) return fail(ISLAM_EARG, "islam_pvgo_eliminate_level0: workspace too small");
    SolvePlan sp;
    if (plan_levels(N, seg_len, sp, solve_twisted()) < 2) return fail(ISLAM_EARG, "islam_pvgo_eliminate_level0: single-level problem");
    Workspace w = carve((void*)align_up((size_t)workspace), N);
    LevelSrc src{};
    src.level0 = 1; src.Hd = Hd; src.Ho = Ho; src.rhs0 = rhs; src.state = nullptr; src.damping_override = damping;
    launch_eliminate(sp.lv[0], sp.twisted != 0, src, level_dst(w.lv[0], w.lv[0].x), w.flags, as_stream(stream), Gate{nullptr, 0.0});
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

// ---- sharded (multi-GPU) building blocks
// The sharded entry points plan like the single-GPU solve (twisted elimination wherever every level qualifies).
static int shard_plan(int N, const int seg_len[2], SolvePlan& sp) { return plan_levels(N, seg_len, sp, solve_twisted()); }

int islam_pvgo_plan(int N, const int seg_len[2], int* plan9) {
    if (N < 1) return fail(ISLAM_EARG, "islam_pvgo_plan: N=%d < 1", N);
    SolvePlan sp;
    const int nl = shard_plan(N, seg_len, sp);
    for (int l = 0; l < MAXL; ++l) {
        plan9[3 * l] = l < nl ? sp.lv[l].n : 0;
        plan9[3 * l + 1] = l < nl ? sp.lv[l].m : 0;
        plan9[3 * l + 2] = l < nl ? sp.lv[l].P : 0;
    }
    plan9[3 * MAXL] = sp.top;
    return nl;
}

static void products_view(double* base, int P, LevelBufs& b) {
    b.Dsep = base; b.rsep = base + 81 * (size_t)P; b.cL = base + 90 * (size_t)P; b.cR = base + 171 * (size_t)P;
    b.fill = base + 252 * (size_t)P; b.cgL = base + 333 * (size_t)P; b.cgR = base + 342 * (size_t)P;
}

// ---- sharding at a HIGHER level of the tree: the interface-only exchange of SURVEY section 8e.
// A rank owns a contiguous range of the segments of the EXCHANGE level xl (the highest level below the root that still has
// one segment per rank) and, below it, everything between the two outer separators of that range: at level l-1 the segments
// whose separators are its level-l nodes.  Levels 0 .. xl are eliminated locally (every block a node needs comes from the
// rank's own segments), only the products of level xl -- 351 doubles per segment, P_xl segments -- are summed over the
// ranks, the few levels above are solved redundantly, and the back-substitution of levels xl .. 0 is local again.
struct ShardRanges { int xl; int seg0[MAXL], nseg[MAXL]; };

static int shard_ranges(const SolvePlan& sp, int world, int rank, ShardRanges& R) {
    if (world < 1 || rank < 0 || rank >= world) return -1;
    int xl = -1;
    for (int l = 0; l < sp.nl - 1; ++l)
        if (sp.lv[l].P >= world) xl = l;
    if (xl < 0) return -1;
    R.xl = xl;
    const int P = sp.lv[xl].P;
    const int s0 = (int)((long long)rank * P / world), s1 = (int)((long long)(rank + 1) * P / world);
    for (int l = 0; l < MAXL; ++l) { R.seg0[l] = 0; R.nseg[l] = 0; }
    R.seg0[xl] = s0;
    R.nseg[xl] = s1 - s0;
    for (int l = xl; l > 0; --l) {
        const int m = sp.lv[l].m, n = sp.lv[l].n;
        const int a = R.seg0[l] * (m + 1);                                                   // first owned node of level l
        const int b = std::min(n - 1, (R.seg0[l] + R.nseg[l] - 1) * (m + 1) + m - 1);       // last one
        R.seg0[l - 1] = a;                                                                    // segment a: right separator = node a
        R.nseg[l - 1] = std::min(b + 1, sp.lv[l - 1].P - 1) - a + 1;                          // ... segment b+1: left separator = node b
        // the last rank owns the chain to its end: when level l closes with a trailing separator (a node of the level above)
        // the level-(l-1) segment beyond it still feeds that separator's block, which the rank's last segment composes
        if (R.seg0[l] + R.nseg[l] == sp.lv[l].P) R.nseg[l - 1] = sp.lv[l - 1].P - a;
    }
    return 0;
}

// The block of an outer separator that a rank hands up (Dsep / rsep of the last segment of each of its levels) is composed
// level by level as  own block - (Schur contribution of the segment on its left) - (that of the segment on its RIGHT); the
// segment on the right belongs to the next rank.  The composition is additive, so the next rank subtracts its share -- the
// left-separator contributions cL / cgL of its first segment of every level below xl -- from the same rows of the exchange
// buffer, and the sum over the ranks is the complete block.
struct OuterFix { const double* cL[MAXL]; const double* cgL[MAXL]; double* Dsep; double* rsep; int n; };
__global__ void outer_block_kernel(OuterFix f, Gate gate) {
    const int t = threadIdx.x;
    if (t >= 90 || gate_closed(gate)) return;
    double v = 0.0;
    for (int i = 0; i < f.n; ++i) v += t < 81 ? f.cL[i][t] : f.cgL[i][t - 81];
    if (t < 81) f.Dsep[t] = -v; else f.rsep[t - 81] = -v;       // a row of the PREVIOUS rank's segment: nothing else is written there locally
}

int islam_pvgo_shard_ranges(int N, const int seg_len[2], int world, int rank, int* out) {
    SolvePlan sp;
    const int nl = shard_plan(N, seg_len, sp);
    ShardRanges R;
    if (nl < 2 || shard_ranges(sp, world, rank, R) != 0)
        return fail(ISLAM_EARG, "islam_pvgo_shard_ranges: N=%d cannot be split over %d ranks (rank %d)", N, world, rank);
    out[0] = R.xl;
    out[1] = sp.lv[R.xl].P;
    for (int l = 0; l < MAXL; ++l) { out[2 + 2 * l] = R.seg0[l]; out[3 + 2 * l] = R.nseg[l]; }
    return ISLAM_OK;
}

static int reproj_dev(const islam_pvgo_reproj* r, ReprojDev& d) {
    if (!r->points || !r->targets || r->K < 1) return fail(ISLAM_EARG, "islam_pvgo_reproj: null points/targets or K=%d < 1", r->K);
    d.points = r->points; d.targets = r->targets; d.K = r->K;
    d.fx = r->fx; d.fy = r->fy; d.cx = r->cx; d.cy = r->cy;
    d.C = {{r->rgb2imu[0], r->rgb2imu[1], r->rgb2imu[2]}, {r->rgb2imu[3], r->rgb2imu[4], r->rgb2imu[5], r->rgb2imu[6]}};
    d.weight = r->weight;
    d.compat_first = r->compat_first_motion;
    return ISLAM_OK;
}

static void enqueue_reproj_reduce(const double* nodes, const double* dx, int M, const ReprojDev& rp, double* red, hipStream_t s,
                                  Gate gate = Gate{nullptr, 0.0}) {
    const int waves = std::min(4, std::max(1, (rp.K + 127) / 128));
    hipLaunchKernelGGL(reproj_reduce_kernel, dim3(xcd_grid(M)), dim3(64 * waves), 64 * waves * (RP_NSUM + 1) * sizeof(double), s, nodes,
                       dx, M, rp, red, gate);
}

// the factor as a rank of the sharded loop sees it: keypoints / targets of its local link 0 = global link `link0`; the frozen
// first motion (pvgo.py:57) belongs to global link 0
static int reproj_dev_local(const islam_pvgo_reproj* r, int link0, ReprojDev& d) {
    const int rc = reproj_dev(r, d);
    if (rc != ISLAM_OK) return rc;
    d.points += (size_t)link0 * d.K * 3;
    d.targets += (size_t)link0 * d.K * 2;
    if (link0 != 0) d.compat_first = 0;
    return ISLAM_OK;
}

static int device_cus() {
    static int cus[64] = {};
    int dev_i = 0;
    if (hipGetDevice(&dev_i) != hipSuccess || dev_i < 0 || dev_i >= 64) return 0;
    if (cus[dev_i] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev_i) != hipSuccess) return 0;
        cus[dev_i] = v;
    }
    return cus[dev_i];
}

// linbuild_kernel / trial_lin_kernel stage their node blocks in dynamic LDS above the default limit (once per device)
static int ensure_linbuild_lds() {
    static bool lb_attr_set[64] = {};                        // per device: the attribute lives in the device's code object
    int dev_i = 0;
    ISLAM_HIP_CHECK(hipGetDevice(&dev_i));
    if (dev_i >= 0 && dev_i < 64 && !lb_attr_set[dev_i]) {
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)linbuild_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LB_DYN_BYTES));
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)trial_lin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LB_DYN_BYTES));
        lb_attr_set[dev_i] = true;
    }
    return ISLAM_OK;
}

}  // extern "C"

#include "pvgo_sharded.inl"   // the sharded LM loop

extern "C" {

int islam_pvgo_shard_upsweep(double* Hd, const double* Ho, const double* rhs, double damping, int N, const int seg_len[2], int world,
                             int rank, int node0, void* workspace, size_t workspace_bytes, double* exchange, int* flags,
                             void* stream) {
    return shard_upsweep_gated(Hd, Ho, rhs, damping, nullptr, N, seg_len, world, rank, node0, workspace, workspace_bytes, exchange, true, flags,
                               Gate{nullptr, 0.0}, as_stream(stream));
}

int islam_pvgo_shard_downsweep(const double* exchange, int N, const int seg_len[2], int world, int rank, int node0, void* workspace,
                               size_t workspace_bytes, double* dx, int* flags, void* stream) {
    return shard_downsweep_gated(exchange, N, seg_len, world, rank, node0, workspace, workspace_bytes, dx, flags, Gate{nullptr, 0.0},
                                 as_stream(stream));
}

// Trial step on M links (nodes/vels/dx hold M+1 rows): writes nodes_t/vels_t (M+1 rows) and part (2 per 64-link block:
// sum r^2 at the trial point, sum JD.(2R+JD)).  Same kernel islam_pvgo_run_chain launches.
int islam_pvgo_trial(const double* nodes, const double* vels, const double* dx, const double* poses, const double* drots,
                     const double* dtrans, const double* dvels, const double* dts, const double* lin, int M, double* nodes_t,
                     double* vels_t, double* part, void* stream) {
    return trial_gated(nodes, vels, dx, poses, drots, dtrans, dvels, dts, lin, M, M, nodes_t, vels_t, part, nullptr, nullptr, nullptr, 0,
                       Gate{nullptr, 0.0}, as_stream(stream));
}

int islam_pvgo_retract(const double* nodes, const double* vels, const double* dx, double sign, int N, double* nodes_out,
                       double* vels_out, void* stream) {
    if (N < 1) return fail(ISLAM_EARG, "islam_pvgo_retract: N=%d < 1", N);
    hipLaunchKernelGGL(retract_kernel, dim3((N + 63) / 64), dim3(64), 0, as_stream(stream), nodes, vels, dx, sign, N,
                       nodes_out, vels_out);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_pvgo_linearize_edges(const double* nodes, const int64_t* edges, const double* poses, int E, double* out,
                               void* stream) {
    if (E < 1) return fail(ISLAM_EARG, "islam_pvgo_linearize_edges: E=%d < 1", E);
    hipLaunchKernelGGL(vo_edge_linearize_kernel, dim3((E + 63) / 64), dim3(64), 0, as_stream(stream), nodes, edges, poses, E, out);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_pvgo_assemble_dense(const double* Hd, const double* Ho, const double* rhs_chain, const double* vo,
                              const int64_t* edges, const int64_t* node_ptr, const int64_t* node_adj, double w0, int N, int E,
                              double* A, double* rhs, void* stream) {
    if (N < 2 || E < 0) return fail(ISLAM_EARG, "islam_pvgo_assemble_dense: N=%d E=%d", N, E);
    hipStream_t s = as_stream(stream);
    ISLAM_HIP_CHECK(hipMemsetAsync(A, 0, (size_t)81 * N * N * sizeof(double), s));
    hipLaunchKernelGGL(dense_nodes_kernel, dim3((N + 63) / 64), dim3(64), 0, s, Hd, Ho, rhs_chain, vo, node_ptr, node_adj, w0, N, E, A,
                       rhs);
    if (E > 0) hipLaunchKernelGGL(dense_edges_kernel, dim3((E + 63) / 64), dim3(64), 0, s, vo, edges, w0, N, E, A);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_pvgo_vo_loss_fwd(const double* nodes, const int64_t* edges, const double* poses, int E, double* err6,
                           double* trans_loss, double* rot_loss, void* stream) {
    if (E < 1) return fail(ISLAM_EARG, "islam_pvgo_vo_loss_fwd: E=%d < 1", E);
    hipLaunchKernelGGL(vo_loss_fwd_kernel, dim3((E + 63) / 64), dim3(64), 0, as_stream(stream), nodes, edges, poses, E,
                       err6, trans_loss, rot_loss);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_pvgo_vo_loss_bwd(const double* poses, const double* err6, const double* g_trans, const double* g_rot, int E,
                           double* grad_poses, void* stream) {
    if (E < 1) return fail(ISLAM_EARG, "islam_pvgo_vo_loss_bwd: E=%d < 1", E);
    hipLaunchKernelGGL(vo_loss_bwd_kernel, dim3((E + 63) / 64), dim3(64), 0, as_stream(stream), poses, err6, g_trans,
                       g_rot, E, grad_poses);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_pvgo_align(const double* nodes, const double* vels, const double* target7, int N, double* nodes_out,
                     double* vels_out, void* stream) {
    if (N < 1) return fail(ISLAM_EARG, "islam_pvgo_align: N=%d < 1", N);
    hipLaunchKernelGGL(align_kernel, dim3((N + 63) / 64), dim3(64), 0, as_stream(stream), nodes, vels, target7, N,
                       nodes_out, vels_out);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_pvgo_reproj_reduce(const double* nodes, const double* dx, int N, const islam_pvgo_reproj* reproj, double* red,
                             void* stream) {
    if (N < 2 || !reproj) return fail(ISLAM_EARG, "islam_pvgo_reproj_reduce: N=%d < 2 or null reproj", N);
    ReprojDev rp{};
    int rc = reproj_dev(reproj, rp);
    if (rc != ISLAM_OK) return rc;
    enqueue_reproj_reduce(nodes, dx, N - 1, rp, red, as_stream(stream));
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_pvgo_run_chain(double* nodes, double* vels, const double* poses, const double* drots, const double* dtrans,
                         const double* dvels, const double* dts, int N, const islam_pvgo_params* prm, void* workspace,
                         size_t workspace_bytes, islam_pvgo_result* result, double* trace, int trace_cap, void* stream) {
    return islam_pvgo_run_chain_reproj(nodes, vels, poses, drots, dtrans, dvels, dts, N, prm, nullptr, workspace,
                                       workspace_bytes, result, trace, trace_cap, stream);
}

}  // extern "C"

// error path of the run: an epoch no enqueued kernel carries turns everything still queued into no-ops
__global__ void close_gate_kernel(double* __restrict__ st) {
    if (threadIdx.x == 0) st[14] = -1.0;
}

static int run_chain_impl(double* nodes, double* vels, const double* poses, const double* drots, const double* dtrans,
                          const double* dvels, const double* dts, int N, const islam_pvgo_params* prm,
                          const islam_pvgo_reproj* reproj, const ReprojDev& rp, Workspace& w, hipStream_t s,
                          islam_pvgo_result* result, double* trace, int trace_cap);

extern "C" {

int islam_pvgo_run_chain_reproj(double* nodes, double* vels, const double* poses, const double* drots, const double* dtrans,
                                const double* dvels, const double* dts, int N, const islam_pvgo_params* prm,
                                const islam_pvgo_reproj* reproj, void* workspace, size_t workspace_bytes,
                                islam_pvgo_result* result, double* trace, int trace_cap, void* stream) {
    if (N < 2) return fail(ISLAM_EARG, "islam_pvgo_run_chain: N=%d < 2", N);
    if (!prm || !result) return fail(ISLAM_EARG, "islam_pvgo_run_chain: null params/result");
    ReprojDev rp{};
    if (reproj) {
        int rc = reproj_dev(reproj, rp);
        if (rc != ISLAM_OK) return rc;
    }
    if (workspace_bytes < islam_pvgo_workspace_bytes(N))
        return fail(ISLAM_EARG, "islam_pvgo_run_chain: workspace %zu < %zu bytes", workspace_bytes,
                    islam_pvgo_workspace_bytes(N));
    Workspace w = carve((void*)align_up((size_t)workspace), N);
    hipStream_t s = as_stream(stream);
    const int rc = run_chain_impl(nodes, vels, poses, drots, dtrans, dvels, dts, N, prm, reproj, rp, w, s, result, trace, trace_cap);
    if (rc != ISLAM_OK) {
        // A failed enqueue or status wait leaves epoch-gated kernels of the run-ahead chain queued: they would still write the pinned
        // status block and the workspace the NEXT call reuses.  Close the gate and drain the stream before handing the error up (the
        // message of the original failure is kept) -- the same rule as the sharded loop (pvgo_dist.hip).
        hipLaunchKernelGGL(close_gate_kernel, dim3(1), dim3(64), 0, s, w.state);
        (void)hipStreamSynchronize(s);
        (void)hipGetLastError();
    }
    return rc;
}

}  // extern "C"

#include "pvgo_lm_loop.inl"   // run_chain_impl

extern "C" {

// Measurement hook (bench.py's roofline leg): the LM loop's dominant launch, trial_elim_kernel, exactly as islam_pvgo_run_chain
// launches it in its steady state -- after one linearisation and one damped solve of the given problem (so that `dx` and the old
// linearisation are real), `launches` back-to-back launches between ONE pair of HIP events on `stream`; *us_per_launch = their
// average period.  info[0..2] = (level-0 segment length, segments, workgroups).  ISLAM_EARG when the fused loop does not apply
// to this problem size (islam_pvgo_run_chain then runs the launch-per-stage loop).
int islam_pvgo_trial_elim_burst(const double* nodes, const double* vels, const double* poses, const double* drots, const double* dtrans,
                                const double* dvels, const double* dts, int N, const islam_pvgo_params* prm, void* workspace,
                                size_t workspace_bytes, int launches, float* us_per_launch, int* info, void* stream) {
    if (N < 2 || !prm || !us_per_launch || launches < 1) return fail(ISLAM_EARG, "islam_pvgo_trial_elim_burst: bad argument");
    if (workspace_bytes < islam_pvgo_workspace_bytes(N)) return fail(ISLAM_EARG, "islam_pvgo_trial_elim_burst: workspace too small");
    SolvePlan sp;
    plan_levels(N, prm->seg_len, sp, solve_twisted());
    const int fz_nwg = std::min(sp.lv[0].P, std::max(device_cus() - 16, 1));
    if (!(sp.twisted && sp.nl >= 2 && sp.top == sp.nl - 1 && sp.lv[0].m <= FZ_MAXM && (sp.lv[0].P + fz_nwg - 1) / fz_nwg <= FZ_S))
        return fail(ISLAM_EARG, "islam_pvgo_trial_elim_burst: the fused loop does not cover N=%d", N);
    Workspace w = carve((void*)align_up((size_t)workspace), N);
    hipStream_t s = as_stream(stream);
    int rc = ensure_linbuild_lds();
    if (rc != ISLAM_OK) return rc;
    ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)trial_elim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FZ_LDS_BYTES));
    ISLAM_HIP_CHECK(hipMemsetAsync(w.ready, 0, w.ready_bytes, s));
    hipLaunchKernelGGL(control_init_kernel, dim3(1), dim3(64), 0, s, w.state, w.flags, prm->radius, prm->down, (uint4*)nullptr, 0u);
    const LinWeights W{prm->w[0], prm->w[1], prm->w[2], prm->w[3], prm->vmin, prm->vmax};
    const int nlb = (N + LB_NODES - 1) / LB_NODES;
    hipLaunchKernelGGL(linbuild_kernel, dim3(xcd_grid(nlb)), dim3(LB_THREADS), LB_DYN_BYTES, s, nodes, vels, poses, drots, dtrans, dvels, dts, N,
                       W, w.lin, w.loss_part, w.Hd, w.Ho, w.rhs, (const double*)nullptr, ReprojDev{}, Gate{nullptr, 0.0});
    hipLaunchKernelGGL(control_begin_kernel, dim3(1), dim3(64), 0, s, w.loss_part, nlb, w.state, w.flags);
    LevelSrc src{};
    src.level0 = 1; src.Hd = w.Hd; src.Ho = w.Ho; src.rhs0 = w.rhs; src.state = w.state; src.hist = 1;
    rc = enqueue_levels(w, sp, 0, src, nullptr, w.dx, w.flags, s, nullptr, nullptr);
    if (rc != ISLAM_OK) return rc;
    FusedArgs fa{};
    fa.nodes = nodes; fa.vels = vels; fa.dx = w.dx; fa.poses = poses; fa.drots = drots; fa.dtrans = dtrans; fa.dvels = dvels; fa.dts = dts;
    fa.lin = w.lin; fa.N = N; fa.nodes_t = w.nodes_t; fa.vels_t = w.vels_t; fa.part = w.part; fa.st = w.state; fa.flags = w.flags;
    fa.ticket = reinterpret_cast<unsigned*>(w.flags + 2);
    fa.tr = TRParams{prm->high, prm->low, prm->up, prm->down, prm->factor, prm->rmin, prm->rmax, prm->reject, prm->max_steps, prm->patience,
                     prm->decreasing};
    fa.report = nullptr; fa.seq = 1.0; fa.W = W;
    fa.lin_o = w.lin2; fa.Hd_o = w.Hd2; fa.Ho_o = w.Ho2; fa.rhs_o = w.rhs2;
    fa.dst = level_dst(w.lv[0], w.dx);
    fa.m = sp.lv[0].m; fa.P = sp.lv[0].P; fa.nwg = fz_nwg;
    fa.eflag = w.flags + 4; fa.eflag_prev = w.flags + 6;
    fa.Ms = N - 1;
    const Gate open{nullptr, 0.0};
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(trial_elim_kernel, dim3(xcd_grid(fz_nwg) + 1), dim3(FZ_THREADS), FZ_LDS_BYTES, s, fa, open);
    hipEvent_t e0, e1;
    ISLAM_HIP_CHECK(hipEventCreate(&e0));
    ISLAM_HIP_CHECK(hipEventCreate(&e1));
    ISLAM_HIP_CHECK(hipStreamSynchronize(s));
    ISLAM_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(trial_elim_kernel, dim3(xcd_grid(fz_nwg) + 1), dim3(FZ_THREADS), FZ_LDS_BYTES, s, fa, open);
    ISLAM_HIP_CHECK(hipEventRecord(e1, s));
    ISLAM_HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0.f;
    ISLAM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *us_per_launch = ms * 1e3f / (float)launches;
    if (info) { info[0] = fa.m; info[1] = fa.P; info[2] = fa.nwg; }
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

}  // extern "C"
