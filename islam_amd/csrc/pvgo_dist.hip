// pvgo_dist.hip -- the LM loop of ONE chain graph sharded over the GPUs of a node, in the library, on RCCL.
//
// No reference counterpart (the reference is single-GPU).  One process per GPU; every rank calls
// islam_pvgo_run_chain_sharded with the SAME full-size inputs (the problem is 0.9 MB) and works on its own stretch of the chain:
//   per LM step     linearise + normal equations of the local links                        (islam_pvgo_linearize / build_normal)
//   per LM trial    up-sweep of the local sub-tree                                          (islam_pvgo_shard_upsweep)
//                   ncclAllReduce #1: the interface blocks, 351 doubles per segment of the exchange level (64.6 KB at
//                                     N = 5001 on 8 ranks)
//                   top levels redundantly + local back-substitution                        (islam_pvgo_shard_downsweep)
//                   trial step on the local links                                           (islam_pvgo_trial)
//                   ncclAllReduce #2: [sum r^2 | sum JD.(2R+JD) | failed pivots | one 10-double halo per rank]
//                   one 24-byte device->host read, then pp.optim.LM's accept / reject rule, TrustRegion.update and
//                   StopOnPlateau on the host -- replicated: every rank sees the same all-reduced scalars.
//   once            ncclAllReduce of the assembled solution (N x 10 doubles, own rows, zero elsewhere).
// The same protocol as islam_amd/dist_pvgo.py (which drives the stage entry points from Python and is what the gloo / virtual-
// rank tests exercise); this file removes the interpreter and torch.distributed from the loop.  comm == NULL: world 1.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.h"

using namespace islam;

namespace {

constexpr int LIN_C = 42;          // components per link of the linearisation buffer (pvgo.hip)

#define ISLAM_NCCL_CHECK(expr)                                                                                    \
    do {                                                                                                          \
        ncclResult_t _r = (expr);                                                                                 \
        if (_r != ncclSuccess) return ::islam::fail(ISLAM_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
    } while (0)

// unweighted loss of the local links [0, n_own): rows of the residuals in the component-major linearisation buffer
__global__ __launch_bounds__(256) void own_loss_kernel(const double* __restrict__ lin, int M, int n_own, double* __restrict__ out) {
    __shared__ double red[256];
    const int rows[15] = {0, 1, 2, 3, 4, 5, 24, 25, 26, 36, 37, 38, 39, 40, 41};
    double s = 0.0;
    for (int k = threadIdx.x; k < n_own; k += 256)
#pragma unroll
        for (int r = 0; r < 15; ++r) { const double v = lin[(size_t)rows[r] * M + k]; s = fma(v, v, s); }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0];
}

// msg = [sum r^2 | sum JD.(2R+JD) | failed | halo of every rank (10 each)]: own part filled, the rest zero
__global__ void pack_msg_kernel(const double* __restrict__ part, int nblk, int* __restrict__ flags, const double* __restrict__ nt,
                                const double* __restrict__ vt, int first_local, int rank, int world, double* __restrict__ msg) {
    const int t = threadIdx.x;
    for (int i = t; i < 3 + 10 * world; i += blockDim.x) msg[i] = 0.0;
    __syncthreads();
    if (t < 2) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += part[2 * b + t];            // fixed order: deterministic
        msg[t] = s;
    }
    if (t == 2) { msg[2] = flags[0] ? 1.0 : 0.0; flags[0] = 0; }
    if (t >= 3 && t < 10) msg[3 + 10 * rank + (t - 3)] = nt[(size_t)first_local * 7 + (t - 3)];
    if (t >= 10 && t < 13) msg[3 + 10 * rank + 7 + (t - 10)] = vt[(size_t)first_local * 3 + (t - 10)];
}

// accepted trial: own rows <- trial rows; the row past the right outer separator <- the next rank's first node (halo)
__global__ __launch_bounds__(256) void accept_kernel(double* __restrict__ nodes, double* __restrict__ vels, const double* __restrict__ nt,
                                                      const double* __restrict__ vt, int rows, const double* __restrict__ halo, int halo_row) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < rows * 7) nodes[i] = nt[i];
    if (i < rows * 3) vels[i] = vt[i];
    if (halo && i < 7) nodes[(size_t)halo_row * 7 + i] = halo[i];
    if (halo && i >= 7 && i < 10) vels[(size_t)halo_row * 3 + (i - 7)] = halo[i];
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

// host-side LM control (islam_amd/lm_control.py; the fused single-GPU loop runs the same rules on the device)
struct Control {
    double radius, damping, high, low, up, down0, down, factor, rmin, rmax, decreasing, loss = 0, last = 0;
    int reject, max_steps, patience, reject_count = 0, steps = 0, patience_count = 0;
    bool continual = true;
    explicit Control(const islam_pvgo_params& p)
        : radius(p.radius), damping(1.0 / p.radius), high(p.high), low(p.low), up(p.up), down0(p.down), down(p.down), factor(p.factor),
          rmin(p.rmin), rmax(p.rmax), decreasing(p.decreasing), reject(p.reject), max_steps(p.max_steps), patience(p.patience) {}
    void begin_step() { last = loss; reject_count = 0; }
    bool after_trial(double loss_trial, double qsum) {
        const double quality = (last - loss_trial) / (-qsum);       // plain IEEE division like PyPose and the device code: 0/0 = NaN -> shrink
        double r = 1.0 / damping;
        if (quality > high) { r = up * r; down = down0; }
        else if (quality > low) { down = down0; }
        else { r = r * down; down = down * factor; }
        down = std::max(rmin, std::min(down, rmax));
        r = std::max(rmin, std::min(r, rmax));
        radius = r; damping = 1.0 / r;
        if (last < loss_trial && reject_count < reject) { loss = last; ++reject_count; return false; }
        loss = loss_trial;
        return true;
    }
    void end_step() {
        ++steps;
        if (steps >= max_steps) continual = false;
        if ((last - loss) < decreasing) ++patience_count; else patience_count = 0;
        if (patience_count >= patience) continual = false;
        if (reject_count >= reject) continual = false;
    }
};

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

size_t islam_pvgo_sharded_scratch_bytes(int N, int world) {
    const size_t n = (size_t)N + 2;
    size_t d = 2 * a256(LIN_C * n) + 2 * a256(81 * n) + 2 * a256(9 * n) + 2 * a256(7 * n) + 2 * a256(3 * n) + a256(2 * (n / 64 + 2)) +
               a256(3 + 10 * (size_t)world) + a256(10 * n) + a256(351 * (n / 5 + 2)) + a256(64);
    return d * sizeof(double) + 512;
}

static int run_sharded(const Reducer& red, int world, int rank, double* nodes, double* vels, const double* poses, const double* drots,
                       const double* dtrans, const double* dvels, const double* dts, int N, const islam_pvgo_params* prm,
                       void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes, islam_pvgo_result* res,
                       long long* exchanged_bytes, void* stream) {
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
    double* lin2 = take(LIN_C * nn);
    double* Hd = take(81 * nn); double* Ho = take(81 * nn); double* rhs = take(9 * nn); double* dx = take(9 * nn);
    double* nl = take(7 * nn); double* nt = take(7 * nn); double* vl = take(3 * nn); double* vt = take(3 * nn);
    double* part = take(2 * (nn / 64 + 2));
    double* msg = take(nmsg);
    double* full = take(10 * nn);
    double* ex = take(351 * (nn / 5 + 2));
    int* flags = (int*)take(64);
    double* lin_own = nullptr;                                     // trial reads the linearisation with the OWN link count as stride
    ISLAM_HIP_CHECK(hipMemsetAsync(flags, 0, 64 * sizeof(double), s));
    ISLAM_HIP_CHECK(hipMemsetAsync(workspace, 0, workspace_bytes, s));     // product rows of other ranks' segments read as zero
    ISLAM_HIP_CHECK(hipMemcpyAsync(nl, nodes + (size_t)a * 7, sizeof(double) * 7 * nloc, hipMemcpyDeviceToDevice, s));
    ISLAM_HIP_CHECK(hipMemcpyAsync(vl, vels + (size_t)a * 3, sizeof(double) * 3 * nloc, hipMemcpyDeviceToDevice, s));
    const double *lp = poses + (size_t)a * 7, *lr = drots + (size_t)a * 4, *ltr = dtrans + (size_t)a * 3, *lv = dvels + (size_t)a * 3,
                 *ldt = dts + a;
    Control ctl(*prm);
    bool has_loss = false;
    int trials = 0, status = ISLAM_OK;
    long long xbytes = 0;
    double host[3];
    if (n_own != Mloc) lin_own = lin2;
    while (ctl.continual) {
        rc = islam_pvgo_linearize(nl, vl, lp, lr, ltr, lv, ldt, nloc, lin, part, s);
        if (rc != ISLAM_OK) return rc;
        if (!has_loss) {
            hipLaunchKernelGGL(own_loss_kernel, dim3(1), dim3(256), 0, s, lin, Mloc, n_own, msg);
            if (world > 1 && (rc = red.sum(msg, 1, s)) != ISLAM_OK) return rc;
            ISLAM_HIP_CHECK(hipMemcpyAsync(host, msg, sizeof(double), hipMemcpyDeviceToHost, s));
            ISLAM_HIP_CHECK(hipStreamSynchronize(s));
            ctl.loss = host[0];
            has_loss = true;
        }
        ISLAM_HIP_CHECK(hipMemsetAsync(Hd, 0, sizeof(double) * 81 * nloc, s));
        ISLAM_HIP_CHECK(hipMemsetAsync(Ho, 0, sizeof(double) * 81 * nloc, s));
        ISLAM_HIP_CHECK(hipMemsetAsync(rhs, 0, sizeof(double) * 9 * nloc, s));
        rc = islam_pvgo_build_normal(lin, ldt, nloc, prm->w, prm->vmin, prm->vmax, Hd, Ho, rhs, s);
        if (rc != ISLAM_OK) return rc;
        if (lin_own)                                                // component-major with stride n_own: 42 strided row copies
            ISLAM_HIP_CHECK(hipMemcpy2DAsync(lin_own, sizeof(double) * n_own, lin, sizeof(double) * Mloc, sizeof(double) * n_own, LIN_C,
                                             hipMemcpyDeviceToDevice, s));
        ctl.begin_step();
        while (true) {
            rc = islam_pvgo_shard_upsweep(Hd, Ho, rhs, ctl.damping, N, prm->seg_len, world, rank, a, workspace, workspace_bytes, ex, flags, s);
            if (rc != ISLAM_OK) return rc;
            if (world > 1 && (rc = red.sum(ex, 351 * (size_t)sh.Pxl, s)) != ISLAM_OK) return rc;
            xbytes += 351LL * sh.Pxl * 8;
            rc = islam_pvgo_shard_downsweep(ex, N, prm->seg_len, world, rank, a, workspace, workspace_bytes, dx, flags, s);
            if (rc != ISLAM_OK) return rc;
            rc = islam_pvgo_trial(nl, vl, dx, lp, lr, ltr, lv, ldt, lin_own ? lin_own : lin, n_own, nt, vt, part, s);
            if (rc != ISLAM_OK) return rc;
            hipLaunchKernelGGL(pack_msg_kernel, dim3(1), dim3(64), 0, s, part, nblk, flags, nt, vt, first_local, rank, world, msg);
            if (world > 1 && (rc = red.sum(msg, nmsg, s)) != ISLAM_OK) return rc;
            xbytes += 8LL * nmsg;
            ISLAM_HIP_CHECK(hipMemcpyAsync(host, msg, 3 * sizeof(double), hipMemcpyDeviceToHost, s));     // the one read of the trial
            ISLAM_HIP_CHECK(hipStreamSynchronize(s));
            ++trials;
            if (host[2] > 0.0) { status = ISLAM_ENOTPD; break; }    // PyPose: "Linear solver failed. Breaking optimization step..."
            if (ctl.after_trial(host[0], host[1])) {
                const bool halo = sh.has_right && rank + 1 < world && b > a + n_own;
                hipLaunchKernelGGL(accept_kernel, dim3((7 * (n_own + 1) + 255) / 256), dim3(256), 0, s, nl, vl, nt, vt, n_own + 1,
                                   halo ? msg + 3 + 10 * (rank + 1) : (const double*)nullptr, b - a);
                break;
            }
        }
        ctl.end_step();
    }
    // the full solution on every rank: own rows (the left outer separator belongs to the previous rank), summed
    ISLAM_HIP_CHECK(hipMemsetAsync(full, 0, sizeof(double) * 10 * (size_t)N, s));
    const int r0 = sh.has_left ? 1 : 0, r1 = n_own + 1;
    hipLaunchKernelGGL(scatter_full_kernel, dim3(((r1 - r0) * 10 + 255) / 256), dim3(256), 0, s, nl, vl, r0, r1, a, full);
    if (world > 1 && (rc = red.sum(full, 10 * (size_t)N, s)) != ISLAM_OK) return rc;
    hipLaunchKernelGGL(unpack_full_kernel, dim3((N * 10 + 255) / 256), dim3(256), 0, s, full, N, nodes, vels);
    ISLAM_LAUNCH_CHECK();
    ISLAM_HIP_CHECK(hipStreamSynchronize(s));
    res->steps = ctl.steps; res->trials = trials; res->status = status; res->loss = ctl.loss; res->damping = ctl.damping;
    if (exchanged_bytes) *exchanged_bytes = xbytes;
    return ISLAM_OK;
}

int islam_pvgo_run_chain_sharded(void* comm, int world, int rank, double* nodes, double* vels, const double* poses, const double* drots,
                                 const double* dtrans, const double* dvels, const double* dts, int N, const islam_pvgo_params* prm,
                                 void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes, islam_pvgo_result* res,
                                 long long* exchanged_bytes, void* stream) {
    Reducer red;
    red.comm = (ncclComm_t)comm;
    return run_sharded(red, world, rank, nodes, vels, poses, drots, dtrans, dvels, dts, N, prm, workspace, workspace_bytes, scratch,
                       scratch_bytes, res, exchanged_bytes, stream);
}

int islam_pvgo_run_chain_sharded_cb(islam_allreduce_fn fn, void* user, int world, int rank, double* nodes, double* vels,
                                    const double* poses, const double* drots, const double* dtrans, const double* dvels, const double* dts,
                                    int N, const islam_pvgo_params* prm, void* workspace, size_t workspace_bytes, void* scratch,
                                    size_t scratch_bytes, islam_pvgo_result* res, long long* exchanged_bytes, void* stream) {
    Reducer red;
    red.fn = fn;
    red.user = user;
    return run_sharded(red, world, rank, nodes, vels, poses, drots, dtrans, dvels, dts, N, prm, workspace, workspace_bytes, scratch,
                       scratch_bytes, res, exchanged_bytes, stream);
}

}  // extern "C"
