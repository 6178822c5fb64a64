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

// ------------------------------------------------------------------------------------------
// residuals of one link (pvgo.py:36-51); also returns what the Jacobian needs
struct LinkRes {
    V3<double> erho, ephi, er, rv, rt;
    SE3<double> pre;      // P^-1 * Xi^-1
    Q4<double> rpre;      // dR^-1 * Ri^-1
};

__device__ __forceinline__ LinkRes link_residuals(SE3<double> Xi, SE3<double> Xj, V3<double> vi, V3<double> vj,
                                                  SE3<double> P, Q4<double> dR, V3<double> dp, V3<double> dv,
                                                  double dt) {
    LinkRes o;
    o.pre = se3_mul(se3_inv(P), se3_inv(Xi));
    se3_log(se3_mul(o.pre, Xj), o.erho, o.ephi);
    o.rv = dv - (vj - vi);
    o.rpre = qmul(qinv(dR), qinv(Xi.q));
    o.er = so3_log(qmul(o.rpre, Xj.q));
    o.rt = (Xj.t - Xi.t) - (dt * vi + dp);
    return o;
}

__device__ __forceinline__ V3<double> ld3(const double* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ Q4<double> ld4(const double* p) { return {p[0], p[1], p[2], p[3]}; }

// one lane per link: residuals + Jacobian blocks G, C (A = [[G, C],[0, G]]) and B
__global__ __launch_bounds__(64) void linearize_kernel(const double* __restrict__ nodes, const double* __restrict__ vels,
                                                        const double* __restrict__ poses, const double* __restrict__ drots,
                                                        const double* __restrict__ dtrans, const double* __restrict__ dvels,
                                                        const double* __restrict__ dts, int M, double* __restrict__ lin,
                                                        double* __restrict__ loss_part) {
    int k = blockIdx.x * 64 + threadIdx.x;
    double sq = 0.0;
    if (k < M) {
        SE3<double> Xi = se3_load(nodes + 7 * k), Xj = se3_load(nodes + 7 * (k + 1));
        LinkRes r = link_residuals(Xi, Xj, ld3(vels + 3 * k), ld3(vels + 3 * (k + 1)), se3_load(poses + 7 * k),
                                   ld4(drots + 4 * k), ld3(dtrans + 3 * k), ld3(dvels + 3 * k), dts[k]);
        // d pgerr / d delta_j = Jl^-1(e) Ad(pre) = [[Ji R, Ji([t]x R - Q Ji R)],[0, Ji R]]
        M3<double> Ji = so3_Jl_inv(r.ephi);
        M3<double> R = qmat(r.pre.q);
        M3<double> G = Ji * R;
        M3<double> C = Ji * (skew(r.pre.t) * R - se3_Q(r.erho, r.ephi) * G);
        M3<double> B = so3_Jl_inv(r.er) * qmat(r.rpre);
        double rec[LIN_C];
        rec[0] = r.erho.x; rec[1] = r.erho.y; rec[2] = r.erho.z;
        rec[3] = r.ephi.x; rec[4] = r.ephi.y; rec[5] = r.ephi.z;
        m3_store(G, rec + 6);
        m3_store(C, rec + 15);
        rec[24] = r.er.x; rec[25] = r.er.y; rec[26] = r.er.z;
        m3_store(B, rec + 27);
        rec[36] = r.rv.x; rec[37] = r.rv.y; rec[38] = r.rv.z;
        rec[39] = r.rt.x; rec[40] = r.rt.y; rec[41] = r.rt.z;
#pragma unroll
        for (int c = 0; c < LIN_C; ++c) lin[(size_t)c * M + k] = rec[c];
        sq = dot(r.erho, r.erho) + dot(r.ephi, r.ephi) + dot(r.rv, r.rv) + dot(r.er, r.er) + dot(r.rt, r.rt);
    }
    sq = wave_sum(sq);
    if (threadIdx.x == 0) loss_part[blockIdx.x] = sq;
}

struct LinkNormal {       // weighted per-link normal-equation pieces
    M3<double> Srr, Srp, Spp;   // pose-pose block S = [[Srr, Srp],[Srp^T, Spp]]
    V3<double> gr, gp;          // J_j^T W r, pose part
    V3<double> rv, rt;
};

__device__ __forceinline__ LinkNormal link_normal(const double* __restrict__ lin, int M, int k, double w0, double w1,
                                                  double w2, double w3) {
    double rec[LIN_C];
#pragma unroll
    for (int c = 0; c < LIN_C; ++c) rec[c] = lin[(size_t)c * M + k];
    V3<double> er{rec[0], rec[1], rec[2]}, ep{rec[3], rec[4], rec[5]}, eR{rec[24], rec[25], rec[26]};
    M3<double> G = m3_load(rec + 6), C = m3_load(rec + 15), B = m3_load(rec + 27);
    M3<double> Gt = transpose(G), Ct = transpose(C), Bt = transpose(B);
    M3<double> GtG = Gt * G;
    LinkNormal o;
    o.rv = {rec[36], rec[37], rec[38]};
    o.rt = {rec[39], rec[40], rec[41]};
    o.Srr = w0 * GtG + w3 * m3_identity<double>();
    o.Srp = w0 * (Gt * C);
    o.Spp = w0 * (Ct * C + GtG) + w2 * (Bt * B);
    o.gr = w0 * (Gt * er) + w3 * o.rt;
    o.gp = w0 * (Ct * er + Gt * ep) + w2 * (Bt * eR);
    return o;
}

__device__ __forceinline__ void put3x3(double* H, int r0, int c0, M3<double> a) {
    H[(r0 + 0) * 9 + c0 + 0] = a.a00; H[(r0 + 0) * 9 + c0 + 1] = a.a01; H[(r0 + 0) * 9 + c0 + 2] = a.a02;
    H[(r0 + 1) * 9 + c0 + 0] = a.a10; H[(r0 + 1) * 9 + c0 + 1] = a.a11; H[(r0 + 1) * 9 + c0 + 2] = a.a12;
    H[(r0 + 2) * 9 + c0 + 0] = a.a20; H[(r0 + 2) * 9 + c0 + 1] = a.a21; H[(r0 + 2) * 9 + c0 + 2] = a.a22;
}

// one lane per node: gather the two adjacent links into Hd[k], Ho[k] (coupling k -> k+1), rhs[k] = -J^T W r
__global__ __launch_bounds__(64) void build_normal_kernel(const double* __restrict__ lin, const double* __restrict__ dts,
                                                           int N, double w0, double w1, double w2, double w3, double vmin,
                                                           double vmax, double* __restrict__ Hd, double* __restrict__ Ho,
                                                           double* __restrict__ rhs) {
    int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= N) return;
    const int M = N - 1;
    const M3<double> Z{0, 0, 0, 0, 0, 0, 0, 0, 0};
    const M3<double> I = m3_identity<double>();
    M3<double> Hrr = Z, Hrp = Z, Hpp = Z;
    V3<double> gr{0, 0, 0}, gp{0, 0, 0}, gv{0, 0, 0};
    double hvv = 0.0, hrv = 0.0;
    if (k > 0) {                       // link k-1, this node is its "j" end
        LinkNormal L = link_normal(lin, M, k - 1, w0, w1, w2, w3);
        Hrr = Hrr + L.Srr; Hrp = Hrp + L.Srp; Hpp = Hpp + L.Spp;
        gr = gr + L.gr; gp = gp + L.gp;
        gv = gv - w1 * L.rv;
        hvv += w1;
    }
    double* o = Ho + (size_t)k * 81;
    if (k < M) {                       // link k, this node is its "i" end
        LinkNormal L = link_normal(lin, M, k, w0, w1, w2, w3);
        double dt = dts[k];
        Hrr = Hrr + L.Srr; Hrp = Hrp + L.Srp; Hpp = Hpp + L.Spp;
        gr = gr - L.gr; gp = gp - L.gp;
        gv = gv + w1 * L.rv - (w3 * dt) * L.rt;
        hvv += w1 + w3 * dt * dt;
        hrv = w3 * dt;
        put3x3(o, 0, 0, -1.0 * L.Srr); put3x3(o, 0, 3, -1.0 * L.Srp); put3x3(o, 0, 6, Z);
        put3x3(o, 3, 0, -1.0 * transpose(L.Srp)); put3x3(o, 3, 3, -1.0 * L.Spp); put3x3(o, 3, 6, Z);
        put3x3(o, 6, 0, (-w3 * dt) * I); put3x3(o, 6, 3, Z); put3x3(o, 6, 6, (-w1) * I);
    }
    double* h = Hd + (size_t)k * 81;
    put3x3(h, 0, 0, Hrr); put3x3(h, 0, 3, Hrp); put3x3(h, 0, 6, hrv * I);
    put3x3(h, 3, 0, transpose(Hrp)); put3x3(h, 3, 3, Hpp); put3x3(h, 3, 6, Z);
    put3x3(h, 6, 0, hrv * I); put3x3(h, 6, 3, Z); put3x3(h, 6, 6, hvv * I);
#pragma unroll
    for (int d = 0; d < 9; ++d) h[d * 10] = fmin(fmax(h[d * 10], vmin), vmax);   // A.diagonal().clamp_(min, max)
    double* b = rhs + (size_t)k * 9;
    b[0] = -gr.x; b[1] = -gr.y; b[2] = -gr.z; b[3] = -gp.x; b[4] = -gp.y; b[5] = -gp.z;
    b[6] = -gv.x; b[7] = -gv.y; b[8] = -gv.z;
}

// ------------------------------------------------------------------------------------------
// Sparse reprojection factor (pvgo.py:53-61 + dense_ba.py:276-305).  Link k: T = C^-1 (X_k^-1 X_{k+1}) C,
// err_j = pixel(K, T^-1 P_j) - target_j.  Under the left perturbation T <- Exp(eta) T:  d p'/d eta = R_T^T [-I, [P]x], so
// with a = (f/z) R^T[row] - (f c/z^2) R^T[2]:  d err / d eta = [-a, a x P].  Node perturbations enter through
// eta = +-Ad(C^-1 X_k^-1) delta (same +/- pattern as the VO factor), applied per link in linbuild / trial.
struct ReprojDev {
    const double* points;
    const double* targets;
    int K;
    double fx, fy, cx, cy;
    SE3<double> C;
    double weight;
    int compat_first;
};
constexpr int RP_REC = ISLAM_REPROJ_REC;    // 21 (J^T J upper) + 6 (J^T r) + 1 (r^T r), padded to 32
constexpr int RP_NSUM = 28;

// one workgroup per link, lanes stride over the keypoints; fixed-order reduction (bit-reproducible)
__global__ __launch_bounds__(256) void reproj_reduce_kernel(const double* __restrict__ nodes, const double* __restrict__ dx,
                                                             int M, ReprojDev rp, double* __restrict__ red, Gate gate) {
    extern __shared__ __attribute__((aligned(16))) double sw[];   // blockDim.x rows of RP_NSUM + 1 doubles
    const int L = xcd_index(blockIdx.x, M);
    if (L < 0 || gate_closed(gate)) return;
    SE3<double> Xi = se3_load(nodes + 7 * L), Xj = se3_load(nodes + 7 * (L + 1));
    if (dx) {
        const double* di = dx + (size_t)L * 9;
        Xi = se3_mul(se3_exp(ld3(di), ld3(di + 3)), Xi);
        Xj = se3_mul(se3_exp(ld3(di + 9), ld3(di + 12)), Xj);
    }
    SE3<double> motion = se3_mul(se3_inv(Xi), Xj);
    const bool frozen = rp.compat_first && L == 0;              // pvgo.py:57: motion[0] = 0.1 (all seven entries)
    if (frozen) motion = {{0.1, 0.1, 0.1}, {0.1, 0.1, 0.1, 0.1}};
    const SE3<double> Tinv = se3_inv(se3_mul(se3_mul(se3_inv(rp.C), motion), rp.C));
    const M3<double> Rm = qmat(Tinv.q);
    const V3<double> r0{Rm.a00, Rm.a01, Rm.a02}, r1{Rm.a10, Rm.a11, Rm.a12}, r2{Rm.a20, Rm.a21, Rm.a22};
    double acc[RP_NSUM];
#pragma unroll
    for (int i = 0; i < RP_NSUM; ++i) acc[i] = 0.0;
    const double* P0 = rp.points + (size_t)L * rp.K * 3;
    const double* T0 = rp.targets + (size_t)L * rp.K * 2;
    for (int j = threadIdx.x; j < rp.K; j += blockDim.x) {
        const V3<double> P = ld3(P0 + 3 * j);
        const V3<double> p = qact(Tinv.q, P) + Tinv.t;
        double den = fmax(fabs(p.z), 2.2250738585072014e-308);     // homo2cart: |z| clamped to finfo.tiny, sign kept
        den = p.z >= 0.0 ? den : -den;
        const double ru = (rp.fx * p.x + rp.cx * p.z) / den - T0[2 * j];
        const double rv = (rp.fy * p.y + rp.cy * p.z) / den - T0[2 * j + 1];
        const double iz = 1.0 / den;
        const V3<double> au = (rp.fx * iz) * r0 - (rp.fx * p.x * iz * iz) * r2;
        const V3<double> av = (rp.fy * iz) * r1 - (rp.fy * p.y * iz * iz) * r2;
        const V3<double> cu = cross(au, P), cv = cross(av, P);
        const double ju[6] = {-au.x, -au.y, -au.z, cu.x, cu.y, cu.z};
        const double jv[6] = {-av.x, -av.y, -av.z, cv.x, cv.y, cv.z};
        int o = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) acc[o++] += ju[a] * ju[b] + jv[a] * jv[b];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] += ju[a] * ru + jv[a] * rv;
        acc[27] += ru * ru + rv * rv;
    }
    // 28 sums over the workgroup through LDS (row stride 29: conflict-free), added in thread order by 28 lanes: a shuffle
    // reduction of a double is two ds_bpermute per step -- 28 x 6 x 2 of them cost more than the keypoint loop
#pragma unroll
    for (int i = 0; i < RP_NSUM; ++i) sw[threadIdx.x * (RP_NSUM + 1) + i] = acc[i];
    __syncthreads();
    if (threadIdx.x < RP_REC) {
        double s = 0.0;
        if (threadIdx.x < RP_NSUM) {
            const int nt = blockDim.x;
            for (int t = 0; t < nt; ++t) s += sw[t * (RP_NSUM + 1) + threadIdx.x];
            if (frozen && threadIdx.x < 27) s = 0.0;            // a constant residual: no Jacobian
        }
        red[(size_t)L * RP_REC + threadIdx.x] = s;
    }
}

// per-link pieces of the reprojection factor in node coordinates: A = M^T S M, g = M^T b with M = Ad(C^-1 X_i^-1)
struct ReprojLink { M3<double> Arr, Arp, App; V3<double> gr, gp; };

__device__ __forceinline__ M3<double> sym_from(const double* u, int r0, int c0) {   // 3x3 sub-block of a packed upper 6x6
    auto at = [&](int r, int c) { if (r > c) { int t = r; r = c; c = t; } return u[r * 6 - r * (r - 1) / 2 + (c - r)]; };
    return {at(r0, c0), at(r0, c0 + 1), at(r0, c0 + 2), at(r0 + 1, c0), at(r0 + 1, c0 + 1), at(r0 + 1, c0 + 2),
            at(r0 + 2, c0), at(r0 + 2, c0 + 1), at(r0 + 2, c0 + 2)};
}

__device__ __forceinline__ void reproj_adjoint(const ReprojDev& rp, SE3<double> Xi, M3<double>& R, M3<double>& T) {
    const SE3<double> Y = se3_mul(se3_inv(rp.C), se3_inv(Xi));
    R = qmat(Y.q);
    T = skew(Y.t) * R;
}

__device__ __forceinline__ ReprojLink reproj_link(const double* __restrict__ rec, const ReprojDev& rp, SE3<double> Xi) {
    double u[RP_NSUM];
#pragma unroll
    for (int i = 0; i < RP_NSUM; ++i) u[i] = rec[i];
    const M3<double> Saa = sym_from(u, 0, 0), Sab = sym_from(u, 0, 3), Sbb = sym_from(u, 3, 3);
    const V3<double> ba{u[21], u[22], u[23]}, bb{u[24], u[25], u[26]};
    M3<double> R, T;
    reproj_adjoint(rp, Xi, R, T);
    const M3<double> Rt = transpose(R), Tt = transpose(T);
    const M3<double> X1 = Saa * T + Sab * R;                    // (S M) top-right
    const M3<double> X2 = transpose(Sab) * T + Sbb * R;         // (S M) bottom-right
    ReprojLink o;
    o.Arr = Rt * (Saa * R);
    o.Arp = Rt * X1;
    o.App = Tt * X1 + Rt * X2;
    o.gr = tmul(R, ba);
    o.gp = tmul(T, ba) + tmul(R, bb);
    return o;
}

// Fused linearise + build (what the LM loop launches): a workgroup of 64 lanes linearises 64 consecutive links (the
// first one is a halo shared with the previous workgroup), hands the weighted per-link pieces over through LDS and builds
// the blocks of its 63 nodes.  Same arithmetic as linearize_kernel + build_normal_kernel, one launch, no re-read of `lin`.
#ifndef ISLAM_LB_NODES
#define ISLAM_LB_NODES 63
#endif
constexpr int LB_NODES = ISLAM_LB_NODES;   // nodes per workgroup of linbuild / trial_lin (<= 63: lane 0 = the link shared with the previous block)
constexpr int LB_THREADS = 256;       // wave 0 linearises the links; waves 0-2 build Hd / Ho / rhs; all four copy out
constexpr int LB_DYN_BYTES = (2 * LB_NODES * 81 + LB_NODES * 9) * (int)sizeof(double);
constexpr int LB_REC = 41;          // Srr 9 | Srp 9 | Spp 9 | gr 3 | gp 3 | rv 3 | rt 3 | dt 1, +1 pad

struct LinWeights { double w0, w1, w2, w3, vmin, vmax; };

// Jacobian blocks of one link at its residuals: d pgerr / d delta_j = [[G, C],[0, G]], d imuroterr / d phi_j = B
__device__ __forceinline__ void link_jacobians(const LinkRes& r, M3<double>& G, M3<double>& C, M3<double>& B) {
    const M3<double> Ji = so3_Jl_inv(r.ephi);
    const M3<double> R = qmat(r.pre.q);
    G = Ji * R;
    C = Ji * (skew(r.pre.t) * R - se3_Q(r.erho, r.ephi) * G);
    B = so3_Jl_inv(r.er) * qmat(r.rpre);
}

// lin record of link L (component-major) + the weighted per-link pieces handed to the node builders through LDS
__device__ __forceinline__ void link_emit(const LinkRes& r, const M3<double>& G, const M3<double>& C, const M3<double>& B,
                                          double dt, int L, int M, bool owns, const LinWeights& W, double* __restrict__ lin,
                                          double* __restrict__ o, const double* __restrict__ red, const ReprojDev& rp,
                                          SE3<double> Xi) {
    if (owns) {                                               // the halo link belongs to the previous workgroup
        double rec[LIN_C];
        rec[0] = r.erho.x; rec[1] = r.erho.y; rec[2] = r.erho.z;
        rec[3] = r.ephi.x; rec[4] = r.ephi.y; rec[5] = r.ephi.z;
        m3_store(G, rec + 6);
        m3_store(C, rec + 15);
        rec[24] = r.er.x; rec[25] = r.er.y; rec[26] = r.er.z;
        m3_store(B, rec + 27);
        rec[36] = r.rv.x; rec[37] = r.rv.y; rec[38] = r.rv.z;
        rec[39] = r.rt.x; rec[40] = r.rt.y; rec[41] = r.rt.z;
#pragma unroll
        for (int c = 0; c < LIN_C; ++c) lin[(size_t)c * M + L] = rec[c];
    }
    const M3<double> Gt = transpose(G), Ct = transpose(C), Bt = transpose(B);
    const M3<double> GtG = Gt * G;
    M3<double> Srr = W.w0 * GtG + W.w3 * m3_identity<double>();
    M3<double> Srp = W.w0 * (Gt * C);
    M3<double> Spp = W.w0 * (Ct * C + GtG) + W.w2 * (Bt * B);
    V3<double> gr = W.w0 * (Gt * r.erho) + W.w3 * r.rt;
    V3<double> gp = W.w0 * (Ct * r.erho + Gt * r.ephi) + W.w2 * (Bt * r.er);
    if (red) {                                                // 5th residual: same +/- coupling pattern as the VO factor
        const ReprojLink q = reproj_link(red + (size_t)L * RP_REC, rp, Xi);
        Srr = Srr + rp.weight * q.Arr; Srp = Srp + rp.weight * q.Arp; Spp = Spp + rp.weight * q.App;
        gr = gr + rp.weight * q.gr; gp = gp + rp.weight * q.gp;
    }
    m3_store(Srr, o);
    m3_store(Srp, o + 9);
    m3_store(Spp, o + 18);
    o[27] = gr.x; o[28] = gr.y; o[29] = gr.z; o[30] = gp.x; o[31] = gp.y; o[32] = gp.z;
    o[33] = r.rv.x; o[34] = r.rv.y; o[35] = r.rv.z; o[36] = r.rt.x; o[37] = r.rt.y; o[38] = r.rt.z; o[39] = dt;
}

// After the link pieces are in `sl` (workgroup barrier done by the caller): waves 0-2 build Hd / Ho / rhs of the
// workgroup's 63 nodes in LDS (node k = links k-1 in slot lane and k in slot lane+1), then all waves copy the three
// contiguous ranges out with lane-contiguous addresses (a lane-per-node store of a 9x9 block touches 64 cache lines).
__device__ __forceinline__ void nodes_build_copy(const double (*sl)[LB_REC], double* __restrict__ lb_out, int blk, int N,
                                                 const LinWeights& W, double* __restrict__ Hd, double* __restrict__ Ho,
                                                 double* __restrict__ rhs) {
    double* const oHd = lb_out;
    double* const oHo = lb_out + LB_NODES * 81;
    double* const oR = lb_out + 2 * LB_NODES * 81;
    const int M = N - 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blk * LB_NODES + lane;
    const int cnt = min(LB_NODES, N - blk * LB_NODES);             // nodes of this workgroup
    const double w1 = W.w1, w3 = W.w3;
    if (lane < cnt && wave < 3) {
        const M3<double> Z{0, 0, 0, 0, 0, 0, 0, 0, 0};
        const M3<double> I = m3_identity<double>();
        const double* a0 = sl[lane];
        const double* a1 = sl[lane + 1];
        if (wave == 0) {                                            // Hd
            M3<double> Hrr = Z, Hrp = Z, Hpp = Z;
            double hvv = 0.0, hrv = 0.0;
            if (k > 0) {
                Hrr = Hrr + m3_load(a0); Hrp = Hrp + m3_load(a0 + 9); Hpp = Hpp + m3_load(a0 + 18);
                hvv += w1;
            }
            if (k < M) {
                const double dt = a1[39];
                Hrr = Hrr + m3_load(a1); Hrp = Hrp + m3_load(a1 + 9); Hpp = Hpp + m3_load(a1 + 18);
                hvv += w1 + w3 * dt * dt;
                hrv = w3 * dt;
            }
            double h[81];
            put3x3(h, 0, 0, Hrr); put3x3(h, 0, 3, Hrp); put3x3(h, 0, 6, hrv * I);
            put3x3(h, 3, 0, transpose(Hrp)); put3x3(h, 3, 3, Hpp); put3x3(h, 3, 6, Z);
            put3x3(h, 6, 0, hrv * I); put3x3(h, 6, 3, Z); put3x3(h, 6, 6, hvv * I);
#pragma unroll
            for (int d = 0; d < 9; ++d) h[d * 10] = fmin(fmax(h[d * 10], W.vmin), W.vmax);   // A.diagonal().clamp_(min, max)
#pragma unroll
            for (int e = 0; e < 81; ++e) oHd[lane * 81 + e] = h[e];
        } else if (wave == 1) {                                     // Ho (coupling k -> k+1); the last node has none
            if (k < M) {
                const M3<double> Srr = m3_load(a1), Srp = m3_load(a1 + 9), Spp = m3_load(a1 + 18);
                const double dt = a1[39];
                double o[81];
                put3x3(o, 0, 0, -1.0 * Srr); put3x3(o, 0, 3, -1.0 * Srp); put3x3(o, 0, 6, Z);
                put3x3(o, 3, 0, -1.0 * transpose(Srp)); put3x3(o, 3, 3, -1.0 * Spp); put3x3(o, 3, 6, Z);
                put3x3(o, 6, 0, (-w3 * dt) * I); put3x3(o, 6, 3, Z); put3x3(o, 6, 6, (-w1) * I);
#pragma unroll
                for (int e = 0; e < 81; ++e) oHo[lane * 81 + e] = o[e];
            }
        } else {                                                    // rhs = -J^T W r
            V3<double> gr{0, 0, 0}, gp{0, 0, 0}, gv{0, 0, 0};
            if (k > 0) {
                gr = gr + ld3(a0 + 27); gp = gp + ld3(a0 + 30);
                gv = gv - w1 * ld3(a0 + 33);
            }
            if (k < M) {
                const double dt = a1[39];
                gr = gr - ld3(a1 + 27); gp = gp - ld3(a1 + 30);
                gv = gv + w1 * ld3(a1 + 33) - (w3 * dt) * ld3(a1 + 36);
            }
            double* bb = oR + lane * 9;
            bb[0] = -gr.x; bb[1] = -gr.y; bb[2] = -gr.z; bb[3] = -gp.x; bb[4] = -gp.y; bb[5] = -gp.z;
            bb[6] = -gv.x; bb[7] = -gv.y; bb[8] = -gv.z;
        }
    }
    __syncthreads();
    const size_t node0 = (size_t)blk * LB_NODES;
    const int nHd = cnt * 81, nHo = min(cnt, M - blk * LB_NODES) * 81, nR = cnt * 9;
    for (int e = threadIdx.x; e < nHd; e += LB_THREADS) Hd[node0 * 81 + e] = oHd[e];
    for (int e = threadIdx.x; e < nHo; e += LB_THREADS) Ho[node0 * 81 + e] = oHo[e];
    for (int e = threadIdx.x; e < nR; e += LB_THREADS) rhs[node0 * 9 + e] = oR[e];
}

__global__ __launch_bounds__(LB_THREADS) void linbuild_kernel(const double* __restrict__ nodes, const double* __restrict__ vels,
                                                               const double* __restrict__ poses, const double* __restrict__ drots,
                                                               const double* __restrict__ dtrans, const double* __restrict__ dvels,
                                                               const double* __restrict__ dts, int N, LinWeights W,
                                                               double* __restrict__ lin, double* __restrict__ loss_part,
                                                               double* __restrict__ Hd, double* __restrict__ Ho,
                                                               double* __restrict__ rhs, const double* __restrict__ red,
                                                               ReprojDev rp, Gate gate) {
    __shared__ double sl[64][LB_REC];
    extern __shared__ __attribute__((aligned(16))) double lb_out[];   // staged Hd (63x81) | Ho (63x81) | rhs (63x9)
    const int M = N - 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int blk = xcd_index(blockIdx.x, (N + LB_NODES - 1) / LB_NODES);
    if (blk < 0 || gate_closed(gate)) return;
    const int L = blk * LB_NODES - 1 + lane;                  // link handled by this lane (wave 0)
    if (wave == 0) {
        double sq = 0.0;
        if (L >= 0 && L < M && lane <= LB_NODES) {
            const SE3<double> Xi = se3_load(nodes + 7 * L), Xj = se3_load(nodes + 7 * (L + 1));
            const double dt = dts[L];
            const LinkRes r = link_residuals(Xi, Xj, ld3(vels + 3 * L), ld3(vels + 3 * (L + 1)), se3_load(poses + 7 * L),
                                             ld4(drots + 4 * L), ld3(dtrans + 3 * L), ld3(dvels + 3 * L), dt);
            M3<double> G, C, B;
            link_jacobians(r, G, C, B);
            const bool owns = lane > 0 || blk == 0;
            if (owns) {
                sq = dot(r.erho, r.erho) + dot(r.ephi, r.ephi) + dot(r.rv, r.rv) + dot(r.er, r.er) + dot(r.rt, r.rt);
                if (red) sq += red[(size_t)L * RP_REC + 27];
            }
            link_emit(r, G, C, B, dt, L, M, owns, W, lin, sl[lane], red, rp, Xi);
        }
        sq = wave_sum(sq);
        if (lane == 0) loss_part[blk] = sq;
    }
    __syncthreads();
    nodes_build_copy(sl, lb_out, blk, N, W, Hd, Ho, rhs);
}

// ------------------------------------------------------------------------------------------
// partitioned block-tridiagonal Cholesky
struct LevelSrc {
    int level0;                  // 1: read Hd/Ho/rhs0 (+ cumulative damping), 0: compose from the previous level
    double* Hd;
    const double* Ho;
    const double* rhs0;
    const double* state;         // state[2] = damping
    double damping_override;     // used when state == nullptr
    int hist;                    // 1: Hd keeps its UNDAMPED diagonal; the dampings of the current linearisation are applied on load
                                 //    (state != nullptr: the list state[16 .. 16 + state[8]]; else damping_override, once)
    const double* zero;          // one double that reads 0.0, in the address space of the arrays above (nullptr: islam_zero16)
    const double *Dsep, *rsep, *cL, *cR, *cgL, *cgR, *fill;
    int Pprev;                   // number of segments of the previous level
};
struct LevelDst {
    double *fac, *inv;           // n x 252, n x 9
    double *Dsep, *rsep;         // (n / stride) x 81, x 9
    double *cL, *cR, *cgL, *cgR, *fill;   // P x 81 / P x 9
    double* x;                   // n x 9, written when the level consists of one segment
};

// Column `lane` of the augmented matrix [S | U | F^T | g] of node k (F^T columns come from elsewhere).
// Per-lane source pointers are fixed for the whole segment (LaneSrc); a node's column is A - B - C
// (level 0: A only).  issue() only loads -- every lane runs the same 9/27 loads, disabled terms read valid
// memory and are dropped by a select in combine() -- so the loads of node c+2 are in flight while node c
// is being eliminated and are first touched one full node later.
// Terms a lane does not take (a lane without a column, the B / C composition terms of a U lane, the coupling of the chain's last
// node, the contribution of a segment that does not exist) are not masked value by value: the lane's POINTER goes to a double that
// reads 0.0 (strides 0), the same loads are issued on every path and combine_cols is a plain a - b - c.  (The masks cost 54
// v_cndmask per node step of the upper levels, a sixth of the sweep's instructions.)
struct LaneSrc {
    const double *A, *B, *C, *Z;
    int nsA, nsB, nsC;        // stride between nodes (doubles)
    int sa, sb, sc;           // stride between rows
    bool isS, isU, isG;
};
struct RawCols { double a[9], b[9], c[9]; };
__device__ double islam_zero16[16];             // never written: reads 0.0

// ZG: the caller provides LevelSrc::zero (sources in LDS: the pointer must stay in that address space)
template <bool ZG = false>
__device__ __forceinline__ LaneSrc lane_source(const LevelSrc& s, int lane) {
    LaneSrc L;
    L.isS = lane < 9; L.isU = lane >= 9 && lane < 18; L.isG = lane == 27;
    L.Z = ZG ? s.zero : islam_zero16;
    const int cu = L.isU ? lane - 9 : 0, cs = L.isS ? lane : 0;
    const bool enA = L.isS || L.isU || L.isG, enBC = L.isS || L.isG;
    L.nsA = enA ? (L.isG ? 9 : 81) : 0;
    L.sa = enA ? (L.isG ? 1 : 9) : 0;
    L.nsB = L.nsC = enBC ? (L.isG ? 9 : 81) : 0;
    L.sb = L.sc = enBC ? (L.isG ? 1 : 9) : 0;
    if (s.level0) {
        L.A = !enA ? L.Z : L.isU ? s.Ho + cu : L.isG ? s.rhs0 : s.Hd + cs;
        L.B = L.C = L.A;
    } else {
        L.A = !enA ? L.Z : L.isU ? s.fill + 81 + cu : L.isG ? s.rsep : s.Dsep + cs;      // U: fill[k+1]
        L.B = !enBC ? L.Z : L.isG ? s.cgR : s.cR + cs;
        L.C = !enBC ? L.Z : L.isG ? s.cgL + 9 : s.cL + 81 + cs;                            // contribution of segment k+1
    }
    return L;
}

// offU: node k has no coupling on this sweep's far side (the chain ends there); offC: there is no segment k+1 to contribute
__device__ __forceinline__ void issue_cols(const LaneSrc& L, bool level0, int k, bool offU, bool offC, RawCols& raw) {
    const bool za = L.isU && offU;
    const double* pa = za ? L.Z : L.A + (size_t)k * L.nsA;
    const int sa = za ? 0 : L.sa;
#pragma unroll
    for (int r = 0; r < 9; ++r) raw.a[r] = pa[r * sa];
    if (!level0) {
        const double* pb = L.B + (size_t)k * L.nsB;
        const double* pc = offC ? L.Z : L.C + (size_t)k * L.nsC;
        const int sc = offC ? 0 : L.sc;
#pragma unroll
        for (int r = 0; r < 9; ++r) raw.b[r] = pb[r * L.sb];
#pragma unroll
        for (int r = 0; r < 9; ++r) raw.c[r] = pc[r * sc];
    }
}

// How the level-0 diagonal is damped (pp.optim.LM: A.diagonal().add_(A.diagonal() * damping), cumulative over the retries of a
// step).  In place: one damping per solve, the damped value is written back for the next retry.  History (LevelSrc::hist): the
// stored diagonal stays undamped and every solve applies the whole list -- same operations in the same order, bit for bit.
struct Damp { double d; const double* list; int n; bool wb; };
__device__ __forceinline__ Damp make_damp(const LevelSrc& s) {
    Damp D;
    D.wb = !s.hist;
    D.list = (s.hist && s.state) ? s.state + STATE_HIST : nullptr;
    D.n = D.list ? (int)s.state[8] + 1 : 0;
    D.d = s.state ? s.state[2] : s.damping_override;
    return D;
}
__device__ __forceinline__ double damp_apply(const Damp& D, double v) {
    if (D.list) {
        for (int i = 0; i < D.n; ++i) v = v + v * D.list[i];
        return v;
    }
    return v + v * D.d;
}

__device__ __forceinline__ void combine_cols(const LaneSrc& L, const LevelSrc& s, int k, int n, int lane, const Damp& damping,
                                             const RawCols& raw, double (&m)[9]) {
    if (s.level0) {
        // lanes that own no column (spike lanes, lanes >= 28) carry don't-care values: they are overwritten by the
        // spike / never stored, so no per-element select is needed except at the chain's last node (no coupling)
        double dg = 0.0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            double v = raw.a[r];
            if (r == lane) { v = damp_apply(damping, v); dg = v; }   // A.diagonal().add_(A.diagonal()*damping), kept for retries
            m[r] = v;
        }
        if (lane < 9 && damping.wb) s.Hd[(size_t)k * 81 + lane * 10] = dg;
    } else {
#pragma unroll
        for (int r = 0; r < 9; ++r) m[r] = raw.a[r] - raw.b[r] - raw.c[r];       // (absent terms were loaded as 0.0: issue_cols)
    }
}

#ifdef ISLAM_PROBE
#define PROBE(slot) do { if (lane == 0 && p == 1 && src.level0) islam_probe_buf[(slot)] = clock64(); } while (0)
#else
#define PROBE(slot) do { } while (0)
#endif

// LDS hand-off inside a one-wave workgroup: LDS instructions of a wave execute in order, so only the
// compiler must be kept from reordering; no vmcnt wait (global prefetches and stores stay in flight).
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// pairs (a <= b) of the symmetric 9x9 accumulation F~^T F~ handled by lanes 0..44, row-major over the upper triangle
// (row a starts at index 9a - a(a-1)/2).  Computed, not tabulated: a table in memory is a dependent load in the prologue
// of every launch, ahead of the first node's column loads in the in-order vmcnt queue.
__device__ __forceinline__ void pair_of(int idx, int& a, int& b) {
    a = (idx >= 9) + (idx >= 17) + (idx >= 24) + (idx >= 30) + (idx >= 35) + (idx >= 39) + (idx >= 42) + (idx >= 44);
    b = idx - (9 * a - ((a * (a - 1)) >> 1)) + a;
}

// LDS column (9 doubles, 16-byte aligned) -> registers with four 16-byte reads and one 8-byte read
__device__ __forceinline__ void ldcol(const double* p, double (&c)[9]) {
    const double2* q = reinterpret_cast<const double2*>(p);
    const double2 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
    c[0] = v0.x; c[1] = v0.y; c[2] = v1.x; c[3] = v1.y; c[4] = v2.x; c[5] = v2.y; c[6] = v3.x; c[7] = v3.y; c[8] = p[8];
}
__device__ __forceinline__ double dot9r(const double (&a)[9], const double (&b)[9]) {
    double s = a[0] * b[0];
#pragma unroll
    for (int q = 1; q < 9; ++q) s = fma(a[q], b[q], s);
    return s;
}

__device__ __forceinline__ double rcp_nr(double p) {      // v_rcp_f64 (~2^-23) + two Newton steps: full double accuracy
    double r = __builtin_amdgcn_rcp(p);
    double e = fma(-p, r, 1.0);
    r = fma(r, e, r);
    e = fma(-p, r, 1.0);
    return fma(r, e, r);
}

// rows of the stored factor of one node needed by lane r: row r of D L^T (9), of U~ (9), of F~ (9), y~[r], 1/p_r
struct FacRow { double lt[9], u[9], f[9], y, iv; };

__device__ __forceinline__ void load_facrow(const double* __restrict__ fac, const double* __restrict__ inv, int c, int r,
                                            FacRow& o) {
    const double* f = fac + (size_t)c * FAC;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.lt[i] = f[i * 9 + r];
#pragma unroll
    for (int q = 0; q < 9; ++q) o.u[q] = f[(9 + q) * 9 + r];
#pragma unroll
    for (int q = 0; q < 9; ++q) o.f[q] = f[(18 + q) * 9 + r];
    o.y = f[27 * 9 + r];
    o.iv = inv[(size_t)c * 9 + r];
}

__device__ __forceinline__ void backsub_run(const double* __restrict__ fac, const double* __restrict__ inv,
                                            double* __restrict__ x, int c0, int cnt, int lane, double (&xn)[9],
                                            const double (&xL)[9], FacRow& cur);

// Back-substitution through one segment (nodes c0 .. c0+cnt-1), right to left; the factor rows of node c-1 are
// fetched while node c is being solved.  xn = solution right of the segment (0 if none), xL = left separator's (0 if none).
__device__ __forceinline__ void backsub_segment(const double* __restrict__ fac, const double* __restrict__ inv,
                                                double* __restrict__ x, int c0, int cnt, int lane, double (&xn)[9],
                                                const double (&xL)[9]) {
    const int r = lane < 9 ? lane : 8;
    FacRow cur;
    load_facrow(fac, inv, c0 + cnt - 1, r, cur);
    backsub_run(fac, inv, x, c0, cnt, lane, xn, xL, cur);
}

// the same with the factor rows of the segment's last node already requested (they do not depend on xn / xL)
__device__ __forceinline__ void backsub_run(const double* __restrict__ fac, const double* __restrict__ inv,
                                            double* __restrict__ x, int c0, int cnt, int lane, double (&xn)[9],
                                            const double (&xL)[9], FacRow& cur) {
    const int r = lane < 9 ? lane : 8;
    FacRow nxt;
    for (int c = c0 + cnt - 1; c >= c0; --c) {
        load_facrow(fac, inv, max(c - 1, c0), r, nxt);       // unconditional (clamped): same memory ops on every path
        __builtin_amdgcn_sched_barrier(0);
        double w = cur.y;
#pragma unroll
        for (int q = 0; q < 9; ++q) w = fma(-cur.u[q], xn[q], w);
#pragma unroll
        for (int q = 0; q < 9; ++q) w = fma(-cur.f[q], xL[q], w);
#pragma unroll
        for (int i = 8; i >= 0; --i) {
            double xi = bcast(w * cur.iv, i);
            xn[i] = xi;
            w = fma(-cur.lt[i], xi, w);      // only rows r < i matter; rows >= i are never read again
        }
        if (lane < 9) {
            double mine = 0.0;
#pragma unroll
            for (int i = 0; i < 9; ++i)
                if (i == lane) mine = xn[i];
            st_coherent(&x[(size_t)c * 9 + lane], mine);
        }
        // the rows of node c-1 were requested at the top of this iteration; make their arrival an explicit event HERE
        // (empty asm with in/out operands), so the next iteration's arithmetic carries no loop-carried memory wait
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            asm volatile("" : "+v"(nxt.lt[i]));
            asm volatile("" : "+v"(nxt.u[i]));
            asm volatile("" : "+v"(nxt.f[i]));
        }
        asm volatile("" : "+v"(nxt.y));
        asm volatile("" : "+v"(nxt.iv));
        cur = nxt;
    }
}

// One wavefront per segment: eliminate the segment's interior nodes onto its two separators (LDL^T, no square roots).
// LDS per wave: Xa = [U- | F- | y-] (rows of L^-1 [U F^T g]), Xb = D^-1 Xa, Tn = Xa^T D^-1 Xa entries for the next node
constexpr int LDS_PER_WAVE = 3 * 19 * XS + XS;       // Xa | Xb | Tn (19 columns each) + one column of zeros behind Tn (twisted_sweep)

// ---- the pivot phase of a node step and the lane map that goes with it.
// ISLAM_PVGO_DPP_PIVOTS (default 1): the multiplier of a row update, element (r, i) of the node's S block, reaches the 28 columns
// through the DP ALU's only DPP form -- `v_fmac_f64_dpp D, D, -f row_newbcast:i`, i.e. m[r] += bcast_i(m[r]) * (-f), ONE instruction
// instead of two v_readlane_b32 + v_fma_f64 (the same operation: a (-f) == (-a) f exactly) -- and the pivot itself through one
// v_mov_b64_dpp.  row_newbcast broadcasts inside a row of 16 lanes, so the nine S columns are held by lanes 16 k + 0..8 of EVERY row
// k (the copies are loaded, damped and updated by the same instructions: they ride along for free) and the 19 other columns
// (U 9..17, spike 18..26, right-hand side 27) by lanes 16 k + 9..15 of rows 0, 1, 2.  The pivot phase of a node step, measured in
// isolation (scripts/probes/pivot_dpp.hip): 948 -> 700 clocks, same bits.  A DPP read needs two wait states after the VALU write of
// its source, which the assembler text cannot leave to the hazard recogniser: s_nop in front of the pivot broadcasts (the row
// updates of a pivot read registers written by the previous pivot's updates, at least the reciprocal's Newton steps earlier).
#ifndef ISLAM_PVGO_DPP_PIVOTS
#define ISLAM_PVGO_DPP_PIVOTS 1
#endif
__device__ __forceinline__ int pivot_col_of(int lane) {
#if ISLAM_PVGO_DPP_PIVOTS
    const int k = lane >> 4, j = lane & 15;
    if (j < 9) return j;
    const int o = 7 * k + (j - 9);
    return o < 19 ? 9 + o : 28;                      // 28: no column (like lanes 28..63 of the identity map)
#else
    return lane;
#endif
}
__device__ __forceinline__ bool pivot_col_primary(int lane) {
#if ISLAM_PVGO_DPP_PIVOTS
    return (lane & 15) < 9 ? lane < 16 : true;
#else
    return true;
#endif
}
#if ISLAM_PVGO_DPP_PIVOTS == 1
template <int I> __device__ __forceinline__ double pivot_bcast(double v) {
    double o;
    if constexpr (I == 0) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
    if constexpr (I == 1) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
    if constexpr (I == 2) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
    if constexpr (I == 3) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
    if constexpr (I == 4) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
    if constexpr (I == 5) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
    if constexpr (I == 6) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
    if constexpr (I == 7) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
    if constexpr (I == 8) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:8 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
    return o;
}
// all row updates of pivot I in ONE assembler block: m[r] += bcast_I(m[r]) * nf for r = I + 1 .. 8.  One block, so that whatever the
// register allocator puts in front of it (copies out of AGPRs under a VGPR cap, PHI copies at the loop head) is followed by the
// block's own two wait states before the first DPP read
template <int I> __device__ __forceinline__ void pivot_updates(double (&m)[9], double nf) {
    if constexpr (I == 0) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %7, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]), "+v"(m[8]) : "v"(nf));
    if constexpr (I == 1) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %6, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]), "+v"(m[8]) : "v"(nf));
    if constexpr (I == 2) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %6 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %6 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %6 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %6 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %6 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %6 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]), "+v"(m[8]) : "v"(nf));
    if constexpr (I == 3) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]), "+v"(m[8]) : "v"(nf));
    if constexpr (I == 4) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "+v"(m[5]), "+v"(m[6]), "+v"(m[7]), "+v"(m[8]) : "v"(nf));
    if constexpr (I == 5) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(m[6]), "+v"(m[7]), "+v"(m[8]) : "v"(nf));
    if constexpr (I == 6) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(m[7]), "+v"(m[8]) : "v"(nf));
    if constexpr (I == 7) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(m[8]) : "v"(nf));
}
template <int I, int N, class F>
__device__ __forceinline__ void pivot_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pivot_static_for<I + 1, N>(f);
    }
}
#endif
// (v_rcp_f64_dpp ASSEMBLES for gfx950 but does not work: scripts/probes/pivot_dpp.hip gets inf from it on every lane -- the DPP operand of
// a double-precision VOP1 instruction is not honoured; only the VOP2 v_fmac_f64 and v_mov_b64 forms are used here.  Also measured and
// dropped: the multiplier as (m r1)(1 + e1) beside the second Newton residual, one operation less on the dependent chain -- 55.9 / 56.3 /
// 56.8 against 55.5 / 55.4 us per LM iteration in alternating runs of two builds on one box.)
// LDL^T pivots of the node's 9x9 S block applied to the lane's column; ipv[i] = 1 / pivot i (every lane), bad |= a non-positive pivot
__device__ __forceinline__ void pivot_phase(double (&mcol)[9], double (&ipv)[9], int& bad) {
#if ISLAM_PVGO_DPP_PIVOTS == 1
    // (nothing that writes a column register may be scheduled into the phase: the assembler text hides its DPP reads from the
    // hazard recogniser, and the wait states in pivot_bcast only cover what was issued before it)
    __builtin_amdgcn_sched_barrier(0);
    pivot_static_for<0, 9>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        const double piv = pivot_bcast<i>(mcol[i]);
        bad |= !(piv > 0.0);                     // off the critical path; a non-positive pivot only poisons this solve
        const double ip = rcp_nr(piv);
        ipv[i] = ip;
        const double nf = -(mcol[i] * ip);
        if constexpr (i < 8) pivot_updates<i>(mcol, nf);
    });
#else
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const double piv = bcast(mcol[i], i);
        bad |= !(piv > 0.0);                     // off the critical path; a non-positive pivot only poisons this solve
        const double ip = rcp_nr(piv);
        ipv[i] = ip;
        const double f = mcol[i] * ip;
#pragma unroll
        for (int r = i + 1; r < 9; ++r) mcol[r] = fma(-bcast(mcol[r], i), f, mcol[r]);
    }
#endif
}

__device__ __forceinline__ void eliminate_segment(const LevelSrc& src, const LevelDst& dst, int n, int m, int p, int* flags,
                                                  int lane, double* __restrict__ lds) {
    double* Xa = lds;
    double* Xb = lds + 19 * XS;
    double* Tn = lds + 2 * 19 * XS;
    const int stride = m + 1;
    const int c0 = p * stride;
    const int cnt = min(m, n - c0);
    const bool has_left = p > 0;
    const int sR = c0 + m;
    const bool has_right = sR < n;
    const Damp damping = make_damp(src);
    const int col = pivot_col_of(lane);                              // the column this lane holds / whether it is the lane that stores it
    const bool prim = pivot_col_primary(lane);                      // (see pivot_col_of)

    // Schur-update work split, fixed per lane: entries (r, cb) = X[:,r] . D^-1 X[:,cb] for cb = g, g+7, g+14
    const int tr = lane % 9, tg = lane / 9;                         // lanes 0..62 (g <= 6); lane 63 idles
    const bool t_on = tg < 7;
    const bool t_third = t_on && (tg + 14) < 19;
    int pa, pb;                                                     // left-separator accumulation F-^T D^-1 [F- | y-]
    pair_of(lane, pa, pb);
    if (lane >= 45) { pa = lane - 45; pb = 9; }
    const bool acc_on = has_left && lane < 54;
    // role of this lane when the next node's columns are formed: S columns and g take (next - update), U columns take
    // the next node's coupling unchanged, spike columns take -(update) (0 without a left separator)
    const bool use_nb = col < 18 || col == 27;
    const bool use_tn = col < 9 || col == 27 || (has_left && col >= 18 && col < 27);
    const int tn_off = (col < 9 ? col : (col >= 18 && col < 27) ? col - 9 : 18) * XS;

    const LaneSrc LS = lane_source(src, col);
    const bool level0 = src.level0 != 0;
    double mcol[9], nb[9];
    RawCols raw;
    issue_cols(LS, level0, c0, (c0 + 1) >= n, (c0 + 1) >= src.Pprev, raw);
    // spike F^T: coupling (left separator rows, c0 cols) transposed; requested together with the first node's columns
    // (every lane loads from a valid address, lanes outside 18..26 / segments without a left separator discard it)
    double spike[9];
    {
        const int jj = (col >= 18 && col < 27) ? col - 18 : 0;
        const int cl = has_left ? c0 : 1;
        const double* O = src.level0 ? (src.Ho + (size_t)(cl - 1) * 81) : (src.fill + (size_t)cl * 81);
#pragma unroll
        for (int r = 0; r < 9; ++r) spike[r] = O[jj * 9 + r];
    }
    combine_cols(LS, src, c0, n, col, damping, raw, mcol);
    if (col >= 18 && col < 27) {
#pragma unroll
        for (int r = 0; r < 9; ++r) mcol[r] = has_left ? spike[r] : 0.0;
    }
    double accL = 0.0;
    int bad = 0;

    PROBE(0);
    for (int t = 0; t < cnt; ++t) {
        const int c = c0 + t;
        const bool last = (t == cnt - 1);
        PROBE(8 * t + 1);
        // prefetch: the columns of node c+1 are requested now and first touched after the whole elimination and Schur
        // update of node c (~1 us later).  Unconditional (index clamped) so that every path through the loop body issues
        // the same memory operations and the compiler can place an exact, late s_waitcnt.
        { const int kn = min(c + 1, n - 1); issue_cols(LS, level0, kn, (kn + 1) >= n, (kn + 1) >= src.Pprev, raw); }
        __builtin_amdgcn_sched_barrier(0);
        // ---- LDL^T elimination of the 9 unknowns of node c, applied to all 28 columns
        double ipv[9];
        pivot_phase(mcol, ipv, bad);
        PROBE(8 * t + 2);
        if (prim && col >= 9 && col < 28) {
            double* xa = Xa + (col - 9) * XS;
            double* xb = Xb + (col - 9) * XS;
#pragma unroll
            for (int r = 0; r < 9; ++r) { xa[r] = mcol[r]; xb[r] = mcol[r] * ipv[r]; }
        }
        lds_sync();
        PROBE(8 * t + 3);
        // ---- Schur update: T = X^T D^-1 X, entries (r, cb), r < 9 (U- columns), cb < 19
        if (t_on) {
            double ca[9], cbv[9];
            ldcol(Xa + tr * XS, ca);
            ldcol(Xb + tg * XS, cbv);
            Tn[tg * XS + tr] = dot9r(ca, cbv);
            ldcol(Xb + (tg + 7) * XS, cbv);
            Tn[(tg + 7) * XS + tr] = dot9r(ca, cbv);
            if (t_third) {
                ldcol(Xb + (tg + 14) * XS, cbv);
                Tn[(tg + 14) * XS + tr] = dot9r(ca, cbv);
            }
        }
        if (acc_on) {
            double ca[9], cbv[9];
            ldcol(Xa + (9 + pa) * XS, ca);
            ldcol(Xb + (9 + pb) * XS, cbv);
            accL += dot9r(ca, cbv);
        }
        lds_sync();
        PROBE(8 * t + 4);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < n) combine_cols(LS, src, c + 1, n, col, damping, raw, nb);
        // the factor goes out only now: vmcnt retires in order, so stores issued before the combine above would have to
        // COMPLETE (write acknowledged, ~0.3 us) before the prefetched columns could be touched
        if (prim && col < 28) {
            double* f = dst.fac + (size_t)c * FAC + col * 9;
#pragma unroll
            for (int r = 0; r < 9; ++r) __builtin_nontemporal_store(mcol[r], &f[r]);   // streamed out: not left dirty in the L2s for the end-of-kernel write-back
        }
        if (lane == 0) {
            double* iv = dst.inv + (size_t)c * 9;
#pragma unroll
            for (int r = 0; r < 9; ++r) iv[r] = ipv[r];
        }
        PROBE(8 * t + 5);
        if (!last) {
            // next node's columns, branch-free: (own column of the next node) - (Schur update column), per-lane role
            double tcol[9];
            ldcol(Tn + tn_off, tcol);
#pragma unroll
            for (int r = 0; r < 9; ++r) mcol[r] = (use_nb ? nb[r] : 0.0) - (use_tn ? tcol[r] : 0.0);
        } else if (has_right) {
            // contributions to the right separator (reduced node p) and the separator's own blocks
            for (int e = lane; e < 81; e += 64) {
                const int r = e / 9, cc = e - r * 9;
                dst.cR[(size_t)p * 81 + e] = Tn[cc * XS + r];
                dst.fill[(size_t)p * 81 + e] = has_left ? -Tn[(9 + r) * XS + cc] : 0.0;   // rows: left sep, cols: right sep
            }
            if (lane < 9) {
                dst.cgR[(size_t)p * 9 + lane] = Tn[18 * XS + lane];
#pragma unroll
                for (int r = 0; r < 9; ++r) dst.Dsep[(size_t)p * 81 + r * 9 + lane] = nb[r];
            }
            if (col == 27 && prim) {
#pragma unroll
                for (int r = 0; r < 9; ++r) dst.rsep[(size_t)p * 9 + r] = nb[r];
            }
        }
        lds_sync();
    }
    if (has_left) {
        if (lane < 45) {
            dst.cL[(size_t)p * 81 + pa * 9 + pb] = accL;
            dst.cL[(size_t)p * 81 + pb * 9 + pa] = accL;
        } else if (lane < 54) {
            dst.cgL[(size_t)p * 9 + (lane - 45)] = accL;
        }
    }
    PROBE(100);
    if (bad && lane == 0) atomicOr(flags, 1);
}

// ------------------------------------------------------------------------------------------
// Twisted (two-sided) elimination of a segment by TWO wavefronts.  Wave A sweeps the interior nodes left -> right exactly
// like eliminate_segment (spike = coupling to the left separator), wave B sweeps right -> left over the mirrored chain
// (its "next node" is c-1, its spike is the coupling to the RIGHT separator); they meet at the middle node c0+h, which A
// eliminates last with the Schur contributions of both sides: a segment of cnt interior nodes costs h+1 = cnt/2+1
// dependent node steps instead of cnt.  The products handed to the next level (cL, cR, fill, cgL, cgR, Dsep, rsep) and
// the factor layout per node are those of eliminate_segment; a B-side node's U~ couples to the node on its LEFT and its
// F~ to the right separator (backsub_twisted).  Segments with fewer than 3 interior nodes run one-sided on wave A.
template <bool ZG = false>
__device__ __forceinline__ LaneSrc lane_source_rev(const LevelSrc& s, int lane) {
    LaneSrc L = lane_source<ZG>(s, lane);
    if (L.isU) {                                    // U' column cu of node k = coupling (k rows, k-1 col cu) = row cu of the
        const int cu = lane - 9;                    // block that couples k-1 -> k
        L.A = (s.level0 ? s.Ho - 81 : s.fill) + cu * 9;
        L.sa = 1;
    }
    return L;
}

__device__ __forceinline__ void combine_cols_rev(const LaneSrc& L, const LevelSrc& s, int k, int lane, const Damp& damping,
                                                 const RawCols& raw, double (&m)[9]) {
    if (s.level0) {
        double dg = 0.0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            double v = raw.a[r];
            if (r == lane) { v = damp_apply(damping, v); dg = v; }
            m[r] = v;
        }
        if (lane < 9 && damping.wb) s.Hd[(size_t)k * 81 + lane * 10] = dg;
    } else {
#pragma unroll
        for (int r = 0; r < 9; ++r) m[r] = raw.a[r] - raw.b[r] - raw.c[r];
    }
}

constexpr int TW_ACC = 9 * 10;                                  // wave B's accumulation onto the right separator: [a][b], b = 9: g
constexpr int LDS_TWISTED = 2 * LDS_PER_WAVE + TW_ACC + 2;      // doubles per workgroup

// Helper wavefronts (HELP = true, bt_eliminate_tw_kernel: four wavefronts per segment).  Of the ~4400 clocks of a node step only
// the pivots, the Schur update and the formation of the next node's columns lie on the k -> k+1 dependency; the factor / reciprocal
// stores (address arithmetic + 18 store instructions per lane) and the accumulation onto the outer separator (two more column
// reads + a 9-term dot product) do not.  Each sweeping wave therefore leaves the eliminated node -- all 28 columns, the D^-1-scaled
// copies and the reciprocal pivots -- in an LDS stage (two stages, alternating with the node's parity) and a helper wave picks it
// up one node step later: it streams the factor out with lane-contiguous 512-byte stores and keeps the separator accumulation.
// One s_barrier per node step (all four waves execute the same number of barriers: the forward sweep's step count) hands a stage
// over; the barrier after step h-1 is also where the forward sweep folds the reverse sweep's side in.
constexpr int H_FST = 28 * XS;                                  // eliminated columns [L^T | U~ | F~ | y~]
constexpr int H_XB = 19 * XS;                                   // D^-1 [U~ | F~ | y~]
constexpr int H_STAGE = H_FST + H_XB + 10;                      // + reciprocal pivots (9, padded)
constexpr int H_SWEEP = 2 * H_STAGE + 20 * XS;                  // two stages + Tn + one column of zeros
constexpr int LDS_TW4 = 2 * H_SWEEP + TW_ACC + 2;               // doubles per workgroup
// PF (upper levels: bt_eliminate_tw_kernel<0>): the helper also FETCHES AND COMPOSES the next node's columns (27 global loads with
// their address arithmetic and the a - b - c per lane -- a sixth of the sweeping wave's instructions, and the sweep is bound by
// instruction issue) and leaves them in LDS, two buffers per sweep by step parity, 29 columns each (28 + one that reads 0.0).
constexpr int H_NB = 29 * XS;
constexpr int LDS_TW4_PF = LDS_TW4 + 4 * H_NB + 2;

// workgroup barrier that waits for this wave's LDS traffic only (no vmcnt wait: global prefetches and stores stay in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One directed sweep.  REV = false: nodes first, first+1, ...; REV = true: first, first-1, ...
//   count      nodes eliminated by this wave
//   has_spike  an outer separator exists (left for forward, right for reverse)
//   merge_t    (forward only) after node step merge_t the next node is the MIDDLE node: wait for wave B and fold its
//              contributions in (-1: one-sided)
//   last_next  (forward only) node whose columns follow the last eliminated node (the right separator), -1: none
//   nbar       (HELP) barriers every wave of the workgroup executes = node steps of the forward sweep
//   L0         1 / 0: the level is known at compile time (level-0 instantiation: no composition loads, fewer registers); -1: runtime
//   PF / nbst  (HELP, upper levels) the next node's columns come composed from the helper: nbst[2][H_NB], buffer t & 1 for step t
template <bool REV, bool HELP = false, int L0 = -1, bool PF = false>
__device__ __forceinline__ void twisted_sweep(const LevelSrc& src_in, const LevelDst& dst, int n, int p, int first, int count,
                                              bool has_spike, int merge_t, int last_next, bool has_right, int* flags, int lane,
                                              double* __restrict__ lds, const double* __restrict__ TnB,
                                              double* __restrict__ accB, const Gate& gate, int nbar = 0,
                                              const double* __restrict__ nbst = nullptr) {
    double* Xa = lds;
    double* Xb = lds + 19 * XS;
    double* Tn = lds + (HELP ? 2 * H_STAGE : 2 * 19 * XS);
    LevelSrc src = src_in;
    if (L0 >= 0) src.level0 = L0 != 0;              // (L0 = 2: level 0 out of LDS blocks, LevelSrc::zero given -- trial_elim_kernel)
    [[maybe_unused]] const bool prb = lane == 0 && p == 1 && !src.level0 && src.Pprev > 500;      // probe build: level 1, segment 1
    [[maybe_unused]] const int pbase = REV ? 470 : 440;
#ifdef ISLAM_PROBE
    const long long t_entry = wall_clock64();
#endif
    const Damp damping = make_damp(src);
    // which of the 28 columns this lane holds (pivot_col_of: with the DP-ALU DPP pivots the nine S columns are replicated in every
    // row of 16 lanes); prim: the lane that stores the column (a replica computes along and stores nothing)
    const int col = pivot_col_of(lane);
    const bool prim = pivot_col_primary(lane);
    const int tr = lane % 9, tg = lane / 9;
    const bool t_on = tg < 7;
    const bool t_third = t_on && (tg + 14) < 19;
    int pa, pb;                                                     // left-separator accumulation F-^T D^-1 [F- | y-]
    pair_of(lane, pa, pb);
    if (lane >= 45) { pa = lane - 45; pb = 9; }
    const bool acc_on = !HELP && has_spike && lane < 54;
    // a lane that takes no Schur-update column (U lanes, lanes without a column, the spike lanes of a sweep without an outer
    // separator) reads the column of zeros behind Tn: the next node's columns are a plain nb - tcol on every lane
    const bool use_tn = col < 9 || col == 27 || (has_spike && col >= 18 && col < 27);
    const int tn_off = (!use_tn ? 19 : col < 9 ? col : (col >= 18 && col < 27) ? col - 9 : 18) * XS;
    if (lane < XS) Tn[19 * XS + lane] = 0.0;
    const LaneSrc LS = REV ? lane_source_rev<L0 == 2>(src, col) : lane_source<L0 == 2>(src, col);
    const bool level0 = src.level0 != 0;
    auto clampi = [&](int k) { return min(max(k, 0), n - 1); };
    double mcol[9], nb[9];
    RawCols raw;
    issue_cols(LS, level0, first, REV ? first <= 0 : (first + 1) >= n, (first + 1) >= src.Pprev, raw);
    double spike[9];
    {
        const int jj = (col >= 18 && col < 27) ? col - 18 : 0;
        if (!REV) {         // coupling (left separator rows, first cols), transposed
            const int cl = has_spike ? first : 1;
            const double* O = src.level0 ? (src.Ho + (size_t)(cl - 1) * 81) : (src.fill + (size_t)cl * 81);
#pragma unroll
            for (int r = 0; r < 9; ++r) spike[r] = O[jj * 9 + r];
        } else {            // coupling (first rows, right separator cols)
            const int cl = first;        // (no right separator: a valid address inside the segment, the value is dropped below --
                                         // row 0 of the chain lies outside a rank's LOCAL level-0 arrays in the sharded solve)
            const double* O = src.level0 ? (src.Ho + (size_t)cl * 81) : (src.fill + (size_t)(cl + 1) * 81);
#pragma unroll
            for (int r = 0; r < 9; ++r) spike[r] = O[r * 9 + jj];
        }
    }
    // the run-ahead gate is looked at only now: the first node's loads are already in flight (a cancelled launch has read
    // valid memory and writes nothing), so the gate word's round trip overlaps them instead of preceding them.  The epoch
    // cannot change while this kernel runs (it is bumped by the previous iteration's trial kernel), so both wavefronts of
    // the workgroup take the same branch.
    if (gate_closed(gate)) return;
#ifdef ISLAM_PROBE
    if (prb) islam_probe_buf[pbase] = t_entry;
#endif
    PROBE_WALL(prb, pbase + 1);
    if (REV) combine_cols_rev(LS, src, first, col, damping, raw, mcol);
    else combine_cols(LS, src, first, n, col, damping, raw, mcol);
    if (col >= 18 && col < 27) {
#pragma unroll
        for (int r = 0; r < 9; ++r) mcol[r] = has_spike ? spike[r] : 0.0;
    }
    double accL = 0.0;
    int bad = 0;
    for (int t = 0; t < count; ++t) {
        const int c = REV ? first - t : first + t;
        const bool last = (t == count - 1);
        PROBE_WALL(prb, pbase + 2 + 5 * t);
        const int nxt = REV ? c - 1 : ((last && last_next >= 0) ? last_next : c + 1);
        if constexpr (!PF) { const int kn = clampi(nxt); issue_cols(LS, level0, kn, REV ? kn <= 0 : (kn + 1) >= n, (kn + 1) >= src.Pprev, raw); }
        __builtin_amdgcn_sched_barrier(0);
        double ipv[9];
        pivot_phase(mcol, ipv, bad);
        PROBE_WALL(prb, pbase + 3 + 5 * t);
        if constexpr (HELP) {
            // the eliminated node goes to the stage of its parity: all 28 columns (the helper streams them out as the factor),
            // the scaled copies of the 19 right-hand columns, the reciprocal pivots
            double* st = lds + (t & 1) * H_STAGE;
            Xa = st + 9 * XS;
            Xb = st + H_FST;
            if (prim && col < 28) {
                double* fc = st + col * XS;
#pragma unroll
                for (int r = 0; r < 9; ++r) fc[r] = mcol[r];
            }
            if (prim && col >= 9 && col < 28) {
                double* xb = Xb + (col - 9) * XS;
#pragma unroll
                for (int r = 0; r < 9; ++r) xb[r] = mcol[r] * ipv[r];
            }
            if (lane < 9) {                      // (lanes 0-8 hold the S columns 0-8 in either lane map)
                double mine = 0.0;
#pragma unroll
                for (int r = 0; r < 9; ++r)
                    if (r == lane) mine = ipv[r];
                st[H_FST + H_XB + lane] = mine;
            }
        } else {
            if (prim && col >= 9 && col < 28) {
                double* xa = Xa + (col - 9) * XS;
                double* xb = Xb + (col - 9) * XS;
#pragma unroll
                for (int r = 0; r < 9; ++r) { xa[r] = mcol[r]; xb[r] = mcol[r] * ipv[r]; }
            }
        }
        lds_sync();
        if (t_on) {
            double ca[9], cbv[9];
            ldcol(Xa + tr * XS, ca);
            ldcol(Xb + tg * XS, cbv);
            Tn[tg * XS + tr] = dot9r(ca, cbv);
            ldcol(Xb + (tg + 7) * XS, cbv);
            Tn[(tg + 7) * XS + tr] = dot9r(ca, cbv);
            if (t_third) {
                ldcol(Xb + (tg + 14) * XS, cbv);
                Tn[(tg + 14) * XS + tr] = dot9r(ca, cbv);
            }
        }
        if (acc_on) {
            double ca[9], cbv[9];
            ldcol(Xa + (9 + pa) * XS, ca);
            ldcol(Xb + (9 + pb) * XS, cbv);
            accL += dot9r(ca, cbv);
        }
        PROBE_WALL(prb, (REV ? 550 : 540) + t);  // (Schur update done, before the barrier)
        if constexpr (HELP) lds_barrier();       // barrier t: stage + Tn complete; the helper takes node c from here
        else lds_sync();
        PROBE_WALL(prb, pbase + 4 + 5 * t);
        __builtin_amdgcn_sched_barrier(0);
        // the next node's own columns; wave B never forms the middle node's (wave A does: its diagonal is damped once)
        const bool want_next = REV ? !last : (nxt >= 0 && nxt < n && (!last || last_next >= 0));
        if (want_next) {
            if constexpr (PF) ldcol(nbst + (t & 1) * H_NB + min(col, 28) * XS, nb);       // composed by the helper before this step's barrier
            else if (REV) combine_cols_rev(LS, src, nxt, col, damping, raw, nb);
            else combine_cols(LS, src, nxt, n, col, damping, raw, nb);
        }
        if constexpr (!HELP) {
            if (prim && col < 28) {
                double* f = dst.fac + (size_t)c * FAC + col * 9;
#pragma unroll
                for (int r = 0; r < 9; ++r) __builtin_nontemporal_store(mcol[r], &f[r]);
            }
            if (lane == 0) {
                double* iv = dst.inv + (size_t)c * 9;
#pragma unroll
                for (int r = 0; r < 9; ++r) iv[r] = ipv[r];
            }
        }
        PROBE_WALL(prb, pbase + 5 + 5 * t);
        if (!last) {
            double tcol[9];
            ldcol(Tn + tn_off, tcol);
#pragma unroll
            for (int r = 0; r < 9; ++r) mcol[r] = nb[r] - tcol[r];       // (nb is 0.0 on lanes without a column of their own: lane_source)
            if (!REV && t == merge_t) {
                // the node just formed is the middle node: fold wave B's side in.  T_B(r, cb): r = middle unknown, cb < 9
                // middle unknown (S update), cb = 9+j right-separator unknown j (its negative IS the coupling middle -> R,
                // i.e. this wave's U columns), cb = 18 right-hand side
                // (HELP: barrier t above is the rendezvous -- the reverse sweep finished its last step before it)
                if constexpr (!HELP) __syncthreads();
                double tb[9];
                const int off = (col < 9 ? col : col < 18 ? col : 18) * XS;           // U column 9+cu reads column 9+cu
                ldcol(TnB + off, tb);
                if (col < 9 || col == 27) {
#pragma unroll
                    for (int r = 0; r < 9; ++r) mcol[r] -= tb[r];
                } else if (col >= 9 && col < 18) {
#pragma unroll
                    for (int r = 0; r < 9; ++r) mcol[r] = has_right ? -tb[r] : 0.0;
                }
            }
        } else if (!REV) {
            if (has_right) {
                const bool addB = merge_t >= 0;               // wave B accumulated onto the right separator as well
                // (HELP: the reverse sweep's helper left that accumulation in accB before it arrived at this step's barrier)
                for (int e = lane; e < 81; e += 64) {
                    const int r = e / 9, cc = e - r * 9;
                    dst.cR[(size_t)p * 81 + e] = Tn[cc * XS + r] + (addB ? accB[r * 10 + cc] : 0.0);
                    dst.fill[(size_t)p * 81 + e] = has_spike ? -Tn[(9 + r) * XS + cc] : 0.0;
                }
                if (lane < 9) {
                    dst.cgR[(size_t)p * 9 + lane] = Tn[18 * XS + lane] + (addB ? accB[lane * 10 + 9] : 0.0);
#pragma unroll
                    for (int r = 0; r < 9; ++r) dst.Dsep[(size_t)p * 81 + r * 9 + lane] = nb[r];
                }
                if (col == 27 && prim) {
#pragma unroll
                    for (int r = 0; r < 9; ++r) dst.rsep[(size_t)p * 9 + r] = nb[r];
                }
            }
        }
        lds_sync();
        PROBE_WALL(prb, pbase + 6 + 5 * t);
    }
    if constexpr (HELP) {
        // the reverse sweep has fewer steps than the forward one: keep the workgroup's barrier count
        for (int t = count; t < nbar; ++t) lds_barrier();
    } else {
        if (!REV) {
            if (has_spike) {
                if (lane < 45) {
                    dst.cL[(size_t)p * 81 + pa * 9 + pb] = accL;
                    dst.cL[(size_t)p * 81 + pb * 9 + pa] = accL;
                } else if (lane < 54) {
                    dst.cgL[(size_t)p * 9 + (lane - 45)] = accL;
                }
            }
        } else {
            // wave B: its last Tn stays in LDS for wave A; the accumulation onto the right separator goes next to it
            if (lane < 45) {
                accB[pa * 10 + pb] = has_spike ? accL : 0.0;
                accB[pb * 10 + pa] = has_spike ? accL : 0.0;
            } else if (lane < 54) {
                accB[(lane - 45) * 10 + 9] = has_spike ? accL : 0.0;
            }
            __syncthreads();
        }
    }
    PROBE_WALL(prb, pbase + 29);
    if (bad && lane == 0) atomicOr(flags, 1);
}

// The helper wavefront (see H_STAGE): after barrier t it owns the stages of node t of both sweeps.  Per node:
//   factor: 252 doubles, lane-contiguous (four 512-byte store instructions instead of nine 72-byte-strided ones per lane)
//   reciprocal pivots: 9 doubles
//   accumulation F~^T D^-1 [F~ | y~] onto the sweep's outer separator (entries as in the sweeping wave: pair_of)
// After the last node it writes cL / cgL (the forward sweep's accumulation onto the left separator); the reverse sweep's sums are
// left in accB BEFORE the workgroup's last barrier (the forward sweep adds them to cR / cgR after it).
__device__ __forceinline__ void helper_node(const LevelDst& dst, int c, const double* __restrict__ st, int lane, bool acc_on, int pa, int pb,
                                            double& accL) {
    double* f = dst.fac + (size_t)c * FAC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        if (e < FAC) {
            const int col = e / 9, row = e - col * 9;
            __builtin_nontemporal_store(st[col * XS + row], &f[e]);
        }
    }
    if (lane < 9) dst.inv[(size_t)c * 9 + lane] = st[H_FST + H_XB + lane];
    if (acc_on) {
        double ca[9], cbv[9];
        ldcol(st + (18 + pa) * XS, ca);                  // F~ column pa
        ldcol(st + H_FST + (9 + pb) * XS, cbv);          // D^-1 [F~ | y~] column pb
        accL += dot9r(ca, cbv);
    }
}

// middle index of a segment with cnt interior nodes (wave A: nodes 0..h incl. the middle, wave B: cnt-1 .. h+1)
__host__ __device__ __forceinline__ int twisted_mid(int cnt) { return cnt >= 3 ? cnt / 2 : cnt - 1; }

// One helper wave can serve several segments of a workgroup (NSEG; trial_elim_kernel: two): nbar = barriers the workgroup executes.
struct HelpSeg { int p, firstA, nA, firstB, nB; bool has_left, has_right, on; const double *ldsA, *ldsB; double* accB;
                 int n, last_next; double *nbA, *nbB; };        // (PF: level size, the node behind the forward sweep's last one, the column buffers)

// PF: composes the columns the sweeps take next (see H_NB): the loads for step t+1 are issued right behind barrier t, the stage of
// node t is streamed out while they fly, the columns are written before barrier t+1.
template <bool REV>
__device__ __forceinline__ bool pf_next(const HelpSeg& g, int t, int& k) {
    int nxt;
    bool want;
    if (REV) { nxt = g.firstB - t - 1; want = t < g.nB - 1; }
    else {
        const bool last = t == g.nA - 1;
        nxt = (last && g.last_next >= 0) ? g.last_next : g.firstA + t + 1;
        want = t < g.nA && nxt < g.n && (!last || g.last_next >= 0);
    }
    k = min(max(nxt, 0), g.n - 1);
    return want;
}

template <int NSEG, bool PF = false>
__device__ __forceinline__ void twisted_helper(const LevelDst& dst, const HelpSeg (&sg)[NSEG], int nbar, int lane,
                                               const LevelSrc* src = nullptr) {
    int pa, pb;
    pair_of(lane, pa, pb);
    if (lane >= 45) { pa = lane - 45; pb = 9; }
    double accA[NSEG], accBv[NSEG];
#pragma unroll
    for (int q = 0; q < NSEG; ++q) { accA[q] = 0.0; accBv[q] = 0.0; }
    [[maybe_unused]] LaneSrc LSA{}, LSB{};
    [[maybe_unused]] RawCols rawA[NSEG], rawB[NSEG];
    [[maybe_unused]] Damp nodamp{0.0, nullptr, 0, false};
    // issue the loads of the columns step t takes next / compose them and leave them in buffer t & 1
    [[maybe_unused]] auto pf_issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < NSEG; ++q) {
            if (!sg[q].on) continue;
            int k;
            if (pf_next<false>(sg[q], t, k)) issue_cols(LSA, false, k, (k + 1) >= sg[q].n, (k + 1) >= src->Pprev, rawA[q]);
            if (pf_next<true>(sg[q], t, k)) issue_cols(LSB, false, k, k <= 0, (k + 1) >= src->Pprev, rawB[q]);
        }
    };
    [[maybe_unused]] auto pf_write = [&](int t) {
#pragma unroll
        for (int q = 0; q < NSEG; ++q) {
            if (!sg[q].on) continue;
            int k;
            double m[9];
            if (pf_next<false>(sg[q], t, k)) {
                combine_cols(LSA, *src, k, sg[q].n, lane, nodamp, rawA[q], m);
                if (lane < 28) {
                    double* o = sg[q].nbA + (t & 1) * H_NB + lane * XS;
#pragma unroll
                    for (int r = 0; r < 9; ++r) o[r] = m[r];
                }
            }
            if (pf_next<true>(sg[q], t, k)) {
                combine_cols_rev(LSB, *src, k, lane, nodamp, rawB[q], m);
                if (lane < 28) {
                    double* o = sg[q].nbB + (t & 1) * H_NB + lane * XS;
#pragma unroll
                    for (int r = 0; r < 9; ++r) o[r] = m[r];
                }
            }
        }
    };
    [[maybe_unused]] const bool hprb = PF && lane == 0 && sg[0].p == 1 && src && src->Pprev > 500;      // probe build: level 1, segment 1
    PROBE_WALL(hprb, 500);
    if constexpr (PF) {
        LSA = lane_source(*src, lane);
        LSB = lane_source_rev(*src, lane);
        pf_issue(0);
#pragma unroll
        for (int q = 0; q < NSEG; ++q) {                     // the column lanes without a column of their own read: 0.0
            if (sg[q].on && lane >= 28 && lane < 28 + XS) {
                sg[q].nbA[28 * XS + lane - 28] = 0.0; sg[q].nbA[H_NB + 28 * XS + lane - 28] = 0.0;
                sg[q].nbB[28 * XS + lane - 28] = 0.0; sg[q].nbB[H_NB + 28 * XS + lane - 28] = 0.0;
            }
        }
        pf_write(0);
    }
    PROBE_WALL(hprb, 501);
    for (int t = 0; t < nbar; ++t) {
#pragma unroll
        for (int q = 0; q < NSEG; ++q) {
            if (sg[q].on && t == sg[q].nA - 1) {             // (nB < nA: the reverse sweep's last node was picked up a step ago)
                double* accB = sg[q].accB;
                if (lane < 45) {
                    accB[pa * 10 + pb] = sg[q].has_right ? accBv[q] : 0.0;
                    accB[pb * 10 + pa] = sg[q].has_right ? accBv[q] : 0.0;
                } else if (lane < 54) {
                    accB[(lane - 45) * 10 + 9] = sg[q].has_right ? accBv[q] : 0.0;
                }
            }
        }
        PROBE_WALL(hprb, 502 + 4 * t);
        lds_barrier();
        PROBE_WALL(hprb, 503 + 4 * t);
        if constexpr (PF) { if (t + 1 < nbar) pf_issue(t + 1); }
#pragma unroll
        for (int q = 0; q < NSEG; ++q) {
            if (!sg[q].on) continue;
            if (t < sg[q].nA) helper_node(dst, sg[q].firstA + t, sg[q].ldsA + (t & 1) * H_STAGE, lane, sg[q].has_left && lane < 54, pa, pb, accA[q]);
            if (t < sg[q].nB) helper_node(dst, sg[q].firstB - t, sg[q].ldsB + (t & 1) * H_STAGE, lane, sg[q].has_right && lane < 54, pa, pb, accBv[q]);
        }
        PROBE_WALL(hprb, 504 + 4 * t);
        if constexpr (PF) { if (t + 1 < nbar) pf_write(t + 1); }
        PROBE_WALL(hprb, 505 + 4 * t);
    }
#pragma unroll
    for (int q = 0; q < NSEG; ++q) {
        if (!sg[q].on || !sg[q].has_left) continue;
        if (lane < 45) {
            dst.cL[(size_t)sg[q].p * 81 + pa * 9 + pb] = accA[q];
            dst.cL[(size_t)sg[q].p * 81 + pb * 9 + pa] = accA[q];
        } else if (lane < 54) {
            dst.cgL[(size_t)sg[q].p * 9 + (lane - 45)] = accA[q];
        }
    }
}

// segment p of a level with n nodes cut into segments of m: what its sweeps and its helper need
struct SegGeom { int c0, cnt, sR, h, nA, nB; bool has_left, has_right, tw; };
__device__ __forceinline__ SegGeom seg_geom(int n, int m, int p) {
    SegGeom g;
    g.c0 = p * (m + 1);
    g.cnt = min(m, n - g.c0);
    g.has_left = p > 0;
    g.sR = g.c0 + m;
    g.has_right = g.sR < n;
    g.tw = g.cnt >= 3;
    g.h = twisted_mid(g.cnt);
    g.nA = g.tw ? g.h + 1 : g.cnt;
    g.nB = g.tw ? g.cnt - 1 - g.h : 0;
    return g;
}
__device__ __forceinline__ HelpSeg help_seg(const SegGeom& g, int p, double* lds_seg, int n = 0) {
    HelpSeg s;
    s.p = p; s.firstA = g.c0; s.nA = g.nA; s.firstB = g.c0 + g.cnt - 1; s.nB = g.nB; s.has_left = g.has_left; s.has_right = g.has_right;
    s.on = true; s.ldsA = lds_seg; s.ldsB = lds_seg + H_SWEEP; s.accB = lds_seg + 2 * H_SWEEP;
    s.n = n; s.last_next = g.has_right ? g.sR : -1; s.nbA = lds_seg + LDS_TW4; s.nbB = lds_seg + LDS_TW4 + 2 * H_NB;
    return s;
}


// uniform: every wave executes exactly one workgroup barrier whatever the segment looks like (several segments share a workgroup)
template <int L0 = -1>
__device__ __forceinline__ void eliminate_twisted(const LevelSrc& src, const LevelDst& dst, int n, int m, int p, int* flags,
                                                  int wave, int lane, double* __restrict__ lds_wg,
                                                  const Gate& gate = Gate{nullptr, 0.0}, bool uniform = false) {
    const int stride = m + 1;
    const int c0 = p * stride;
    const int cnt = min(m, n - c0);
    const bool has_left = p > 0;
    const int sR = c0 + m;
    const bool has_right = sR < n;
    const bool tw = cnt >= 3;
    const int h = twisted_mid(cnt);
    double* ldsA = lds_wg;
    double* ldsB = lds_wg + LDS_PER_WAVE;
    double* accB = lds_wg + 2 * LDS_PER_WAVE;
    if (wave == 0) {
        twisted_sweep<false, false, L0>(src, dst, n, p, c0, tw ? h + 1 : cnt, has_left, tw ? h - 1 : -1, has_right ? sR : -1, has_right, flags,
                                        lane, ldsA, ldsB + 2 * 19 * XS, accB, gate);
        if (uniform && !tw) __syncthreads();
    } else if (tw) {
        twisted_sweep<true, false, L0>(src, dst, n, p, c0 + cnt - 1, cnt - 1 - h, has_right, -1, -1, has_right, flags, lane, ldsB, nullptr, accB, gate);
    } else if (uniform) {
        __syncthreads();
    }
}

// The two sweeps of segment p with helper hand-off (HELP): role 0 = forward, 1 = reverse.  nbar = barriers every wave of the
// workgroup executes (>= this segment's forward step count; more when a workgroup holds segments of different lengths).
template <int L0, bool PF = false>
__device__ __forceinline__ void sweep_with_helper(const LevelSrc& src, const LevelDst& dst, int n, int m, int p, int* flags, int role, int lane,
                                                  double* __restrict__ lds_seg, const Gate& gate, int nbar) {
    const SegGeom g = seg_geom(n, m, p);
    double* ldsA = lds_seg;
    double* ldsB = lds_seg + H_SWEEP;
    double* accB = lds_seg + 2 * H_SWEEP;
    if (role == 0)
        twisted_sweep<false, true, L0, PF>(src, dst, n, p, g.c0, g.nA, g.has_left, g.tw ? g.h - 1 : -1, g.has_right ? g.sR : -1, g.has_right,
                                           flags, lane, ldsA, ldsB + 2 * H_STAGE, accB, gate, nbar, lds_seg + LDS_TW4);
    else if (g.tw)
        twisted_sweep<true, true, L0, PF>(src, dst, n, p, g.c0 + g.cnt - 1, g.nB, g.has_right, -1, -1, g.has_right, flags, lane, ldsB, nullptr,
                                          accB, gate, nbar, lds_seg + LDS_TW4 + 2 * H_NB);
    else if (!gate_closed(gate)) { for (int t = 0; t < nbar; ++t) lds_barrier(); }
}

// three wavefronts per segment: 0 = forward sweep, 1 = reverse sweep, 2 = the helper of both
template <int L0>
__device__ __forceinline__ void eliminate_twisted3(const LevelSrc& src, const LevelDst& dst, int n, int m, int p, int* flags,
                                                   int wave, int lane, double* __restrict__ lds_wg, const Gate& gate) {
    const SegGeom g = seg_geom(n, m, p);
    constexpr bool PF = L0 == 0;                     // upper levels: the helper fetches and composes the next node's columns
    if (wave < 2) sweep_with_helper<L0, PF>(src, dst, n, m, p, flags, wave, lane, lds_wg, gate, g.nA);
    else if (!gate_closed(gate)) {
        const HelpSeg sg[1] = {help_seg(g, p, lds_wg, n)};
        twisted_helper<1, PF>(dst, sg, g.nA, lane, &src);
    }
}

// (level 0 of the N = 5001 tree has 834 segments, all of which must be resident at once: 3 waves per SIMD, i.e. <= 168 VGPRs)
template <int L0>
__global__ __launch_bounds__(192, L0 ? 3 : 2) void bt_eliminate_tw_kernel(LevelSrc src, LevelDst dst, int n, int m, int* flags, int seg0,
                                                                          int nseg, Gate gate) {
    __shared__ __attribute__((aligned(16))) double lds[L0 ? LDS_TW4 : LDS_TW4_PF];
    const int p = xcd_index(blockIdx.x, nseg);
    if (p < 0) return;
    eliminate_twisted3<L0>(src, dst, n, m, p + seg0, flags, threadIdx.x >> 6, threadIdx.x & 63, lds, gate);
}

#ifdef ISLAM_PROBE
extern "C" int islam_probe_read(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(islam_probe_buf), sizeof(long long) * 1024) == hipSuccess ? 0 : -2;
}
#endif

// one wavefront per workgroup, one segment per workgroup (the large levels)
__global__ __launch_bounds__(64) void bt_eliminate_kernel(LevelSrc src, LevelDst dst, int n, int m, int* flags, int seg0,
                                                           int nseg, Gate gate) {
    __shared__ __attribute__((aligned(16))) double lds[LDS_PER_WAVE];
    const int p = xcd_index(blockIdx.x, nseg);
    if (p < 0 || gate_closed(gate)) return;
    eliminate_segment(src, dst, n, m, p + seg0, flags, threadIdx.x, lds);
}

// Segment of at most BS_PAR_MAX nodes: lane t*9 + r owns row r of node c0+t, so the factor rows of ALL nodes are requested
// at once (one memory round trip instead of one per node) and the left-separator term y - F~ xL of every node is formed in
// parallel; only U~ x_{t+1} and the 9-step triangular solve remain sequential (pivot lane broadcast with v_readlane).
constexpr int BS_PAR_MAX = 7;

__device__ __forceinline__ void backsub_par_load(const double* __restrict__ fac, const double* __restrict__ inv, int c0,
                                                 int cnt, int lane, FacRow& row) {
    const int t = min(lane / 9, cnt - 1);           // lanes past the segment duplicate its last node (valid memory, unused)
    load_facrow(fac, inv, c0 + t, lane - (lane / 9) * 9, row);
}

__device__ __forceinline__ void backsub_par_run(double* __restrict__ x, int c0, int cnt, int lane, double (&xn)[9],
                                                const double (&xL)[9], const FacRow& row) {
    const int t = lane / 9, r = lane - t * 9;
    double wF = row.y;
#pragma unroll
    for (int q = 0; q < 9; ++q) wF = fma(-row.f[q], xL[q], wF);
    for (int tt = cnt - 1; tt >= 0; --tt) {
        double w = wF;                                // meaningful on the nine lanes of node tt
#pragma unroll
        for (int q = 0; q < 9; ++q) w = fma(-row.u[q], xn[q], w);
        const int base = tt * 9;
#pragma unroll
        for (int i = 8; i >= 0; --i) {
            const double xi = bcast(w * row.iv, base + i);
            xn[i] = xi;
            w = fma(-row.lt[i], xi, w);
        }
        if (t == tt) {
            double mine = 0.0;
#pragma unroll
            for (int i = 0; i < 9; ++i)
                if (i == r) mine = xn[i];
            st_coherent(&x[(size_t)(c0 + tt) * 9 + r], mine);
        }
    }
}

// Back-substitution of a segment factored by eliminate_twisted: the middle node first (its U~ couples to the right
// separator), then both halves at once -- the nodes left of the middle right-to-left (U~ couples to the node on the right,
// F~ to the left separator) and the nodes right of it left-to-right (U~ couples to the node on the LEFT, F~ to the right
// separator): h+1 dependent node steps instead of cnt.  xR / xL = solution at the right / left separator (0 if none).
__device__ __forceinline__ void backsub_par_run_tw(double* __restrict__ x, int c0, int cnt, int lane, const double (&xR)[9],
                                                   const double (&xL)[9], const FacRow& row) {
    const int t = lane / 9, r = lane - t * 9;
    const int h = twisted_mid(cnt);
    const bool sideB = t > h;
    double wF = row.y;
#pragma unroll
    for (int q = 0; q < 9; ++q) wF = fma(-row.f[q], sideB ? xR[q] : xL[q], wF);
    double xn[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) xn[q] = xR[q];
    {
        double w = wF;
#pragma unroll
        for (int q = 0; q < 9; ++q) w = fma(-row.u[q], xn[q], w);
        const int base = h * 9;
#pragma unroll
        for (int i = 8; i >= 0; --i) {
            const double xi = bcast(w * row.iv, base + i);
            xn[i] = xi;
            w = fma(-row.lt[i], xi, w);
        }
        if (t == h) {
            double mine = 0.0;
#pragma unroll
            for (int i = 0; i < 9; ++i)
                if (i == r) mine = xn[i];
            st_coherent(&x[(size_t)(c0 + h) * 9 + r], mine);
        }
    }
    const int steps = max(h, cnt - 1 - h);
    for (int j = 1; j <= steps; ++j) {
        const int tA = h - j, tB = h + j;
        double w = wF;
#pragma unroll
        for (int q = 0; q < 9; ++q) w = fma(-row.u[q], xn[q], w);
        const int baseA = max(tA, 0) * 9, baseB = min(tB, cnt - 1) * 9;
#pragma unroll
        for (int i = 8; i >= 0; --i) {
            const double v = w * row.iv;
            const double xa = bcast(v, baseA + i);
            const double xb = bcast(v, baseB + i);
            const double xi = sideB ? xb : xa;
            xn[i] = xi;
            w = fma(-row.lt[i], xi, w);
        }
        if ((t == tA && tA >= 0) || (t == tB && tB < cnt)) {
            double mine = 0.0;
#pragma unroll
            for (int i = 0; i < 9; ++i)
                if (i == r) mine = xn[i];
            st_coherent(&x[(size_t)(c0 + t) * 9 + r], mine);
        }
    }
}

// The back-substitution of a twisted segment is LINEAR in the two separator solutions:  x_t = G_t [1; -xL; -xR]  with a 9 x 19
// matrix G_t per interior node that depends on the factor only.  influence_tw computes the G_t BEFORE the separators are known
// (the down-sweep's workgroups wait 7-14 us for them anyway, their factor rows already loaded): the same recurrences as
// backsub_par_run_tw with 19 right-hand sides -- [y | F~ | U~] for the middle node, [y | F~ | 0] - U~ G_{t+1} left of it,
// [y | 0 | F~] - U~ G_{t-1} right of it -- one lane per column, the two halves of the segment on the two halves of the wave, the
// factor rows broadcast from LDS.  What is left on the critical path once the separators arrive is one 18-term dot product per
// lane instead of h+1 dependent 9-step triangular solves (1.45 -> ~0.2 us per level of the tree).
// In: lane 9t+r holds FacRow `row` of node t (t < cnt <= BS_PAR_MAX).  Out: g = row r of G_t on lane 9t+r.
constexpr int INF_FR = 22;                                  // doubles per (node, row) record in LDS: u 9 (+1) | f 9 (+1) | y | pad: 16-byte aligned pieces
constexpr int INF_ND = 9 * 10 + 10;                         // per node: the transposed L^T block (column i = 9 doubles, stride 10) | reciprocal pivots 9 (+1)
constexpr int INF_GS = 20;                                  // row stride of G in LDS (19 + pad)
constexpr int LDS_INFLUENCE = 9 * BS_PAR_MAX * (INF_FR + INF_GS) + BS_PAR_MAX * INF_ND;
__device__ __forceinline__ void influence_tw(const FacRow& row, int cnt, int lane, double* __restrict__ lds, double (&g)[19]) {
    double* rec = lds;                                      // [9 * cnt][INF_FR]
    double* G = lds + 9 * BS_PAR_MAX * INF_FR;              // [9 * cnt][INF_GS]
    double* nd = G + 9 * BS_PAR_MAX * INF_GS;               // [cnt][INF_ND]
    // Layout for wide, mostly broadcast reads (the recurrences below are one wavefront's chain of LDS round trips and fp64 FMAs):
    // a row's U~ and F~ are 16-byte aligned runs of nine, the entries of D L^T a triangular-solve step needs -- column i above the
    // diagonal -- are contiguous in a transposed copy per node, and so are the node's reciprocal pivots.
    if (lane < 9 * cnt) {
        double* p = rec + lane * INF_FR;
        const int t = lane / 9, r = lane - 9 * t;
        double* n = nd + t * INF_ND;
#pragma unroll
        for (int q = 0; q < 9; ++q) { p[q] = row.u[q]; p[10 + q] = row.f[q]; n[q * 10 + r] = row.lt[q]; }
        p[20] = row.y;
        n[90 + r] = row.iv;
    }
    lds_sync();
    const int h = twisted_mid(cnt);
    const int grp = lane >> 5, c = lane & 31;               // group 0: middle, h-1, ..., 0; group 1: (middle,) h+1, ..., cnt-1
    const bool col = c < 19;
    const int cc = col ? c : 0;
    // right-hand side column c of node tn: kind 0 = middle, 1 = left of it (side A), 2 = right of it (side B)
    auto solve = [&](int tn, int kind, double (&X)[9]) {
        const double* R = rec + (size_t)tn * 9 * INF_FR;
        const double* n = nd + (size_t)tn * INF_ND;
        double b[9];
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
            const double* p = R + rr * INF_FR;
            double v = 0.0;
            if (cc == 0) v = p[20];
            else if (cc < 10) v = kind != 2 ? p[10 + cc - 1] : 0.0;                      // F~ multiplies xL for the middle and side A
            else v = kind == 0 ? p[cc - 10] : (kind == 2 ? p[10 + cc - 10] : 0.0);       // middle: U~ multiplies xR; side B: F~ does
            if (kind != 0) {
                double u[9];
                ldcol(p, u);
#pragma unroll
                for (int q = 0; q < 9; ++q) v = fma(-u[q], X[q], v);                     // - U~ G_neighbour
            }
            b[rr] = v;
        }
        double iv[9];
        ldcol(n + 90, iv);
#pragma unroll
        for (int i = 8; i >= 0; --i) {
            const double xi = b[i] * iv[i];
            X[i] = xi;
            if (i > 0) {
                double lc[9];                                // column i of D L^T: rows 0 .. i-1 matter
                ldcol(n + i * 10, lc);
#pragma unroll
                for (int rr = 0; rr < 9; ++rr)
                    if (rr < i) b[rr] = fma(-lc[rr], xi, b[rr]);
            }
        }
    };
    double X[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) X[q] = 0.0;
    solve(h, 0, X);                                         // both groups: each chain starts from the middle node's G
    if (grp == 0 && col) {
#pragma unroll
        for (int q = 0; q < 9; ++q) G[(h * 9 + q) * INF_GS + cc] = X[q];
    }
    const int steps = max(h, cnt - 1 - h);
    for (int j = 1; j <= steps; ++j) {
        const int tn = grp == 0 ? h - j : h + j;
        const bool on = grp == 0 ? tn >= 0 : tn < cnt;
        const int tc = min(max(tn, 0), cnt - 1);            // inactive lanes recompute a valid node and drop the result
        double Y[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) Y[q] = X[q];
        solve(tc, grp == 0 ? 1 : 2, Y);
        if (on) {
#pragma unroll
            for (int q = 0; q < 9; ++q) X[q] = Y[q];
            if (col) {
#pragma unroll
                for (int q = 0; q < 9; ++q) G[(tc * 9 + q) * INF_GS + cc] = Y[q];
            }
        }
    }
    lds_sync();
    const int lr = min(lane, 9 * cnt - 1);
#pragma unroll
    for (int k = 0; k < 19; ++k) g[k] = G[lr * INF_GS + k];
    lds_sync();
}

__device__ __forceinline__ void backsub_level_segment(const double* __restrict__ fac, const double* __restrict__ inv,
                                                      const double* __restrict__ xsep, double* __restrict__ x, int n, int m,
                                                      int p, int lane) {
    const int stride = m + 1;
    const int c0 = p * stride;
    const int cnt = min(m, n - c0);
    const bool has_left = p > 0;
    const int sR = c0 + m;
    const bool has_right = sR < n;
    double xn[9], xL[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        xL[q] = has_left ? xsep[(size_t)(p - 1) * 9 + q] : 0.0;
        xn[q] = has_right ? xsep[(size_t)p * 9 + q] : 0.0;
    }
    if (has_right && lane < 9) x[(size_t)sR * 9 + lane] = xsep[(size_t)p * 9 + lane];
    if (m <= BS_PAR_MAX) {
        FacRow row;
        backsub_par_load(fac, inv, c0, cnt, lane, row);
        backsub_par_run(x, c0, cnt, lane, xn, xL, row);
    } else {
        backsub_segment(fac, inv, x, c0, cnt, lane, xn, xL);
    }
}

// The small top of the level tree in ONE launch: a single workgroup of up to 8 wavefronts runs every remaining level
// (wave w = segment w), separated by workgroup barriers instead of kernel boundaries: up-sweep, root solve, down-sweep.
constexpr int MAXTOP = 4;
struct TopArgs {
    LevelSrc src[MAXTOP];
    LevelDst dst[MAXTOP];
    int n[MAXTOP], m[MAXTOP], P[MAXTOP];
    int nl;
};

__global__ __launch_bounds__(512) void bt_top_kernel(TopArgs a, int* flags, Gate gate) {
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    if (gate_closed(gate)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* lds = lds_all + wave * LDS_PER_WAVE;
    for (int l = 0; l < a.nl; ++l) {
        if (wave < a.P[l]) eliminate_segment(a.src[l], a.dst[l], a.n[l], a.m[l], wave, flags, lane, lds);
        __syncthreads();                       // the level's products are visible to the whole workgroup
    }
    const int top = a.nl - 1;                  // P[top] == 1: the root level is fully factored by wave 0
    if (wave == 0) {
        double xn[9], xL[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) { xn[q] = 0.0; xL[q] = 0.0; }
        backsub_segment(a.dst[top].fac, a.dst[top].inv, a.dst[top].x, 0, a.n[top], lane, xn, xL);
    }
    for (int l = top - 1; l >= 0; --l) {
        __syncthreads();
        if (wave < a.P[l]) backsub_level_segment(a.dst[l].fac, a.dst[l].inv, a.dst[l + 1].x, a.dst[l].x, a.n[l], a.m[l], wave, lane);
    }
}

// expand the solution of the separators (xsep, from the next level) into this level's interior nodes
__global__ __launch_bounds__(64) void bt_backsub_kernel(const double* __restrict__ fac, const double* __restrict__ inv,
                                                         const double* __restrict__ xsep, double* __restrict__ x, int n,
                                                         int m, int seg0, int nseg, Gate gate) {
    const int p = xcd_index(blockIdx.x, nseg);
    if (p < 0 || gate_closed(gate)) return;
    backsub_level_segment(fac, inv, xsep, x, n, m, p + seg0, threadIdx.x);
}

// Root solve + the whole down-sweep in ONE launch.  Workgroup 0 eliminates and solves the root level; every other
// workgroup owns one segment of one level (upper levels first in workgroup order).  A segment requests the factor rows
// of its last node (they do not depend on the solution above), waits until the one or two segments of the level above
// that hold its separators have published their part of the solution (a per-segment word set to this solve's serial
// number after an agent-scope release fence), back-substitutes, publishes.  The waits replace four kernel boundaries and
// overlap the first factor loads with the dependency.  Workgroups are dispatched in index order, so a waiting workgroup
// only ever waits for lower-indexed ones (already resident or finished); the spin is bounded all the same.
// seg0 / nseg: the window of segments this launch back-substitutes (all P of them on one GPU; a rank's own range in the sharded
// solve).  outer: the window's first segment takes its LEFT separator -- the rank's left cut node, a node of every level up to
// the replicated ones -- from SweepArgs::outer_x instead of the level above (whose segment holding it belongs to the
// previous rank).  x_last: last valid index of x (the sharded solve hands in a local array).
// merge: this level does not wait for the level above it (the PRODUCER, lv[i-1]) but composes its influence matrices with the
// producer's -- published through gx / the G-ready words gflag0 + segment -- and takes its solution straight from the level
// above the producer.  publish_g: this level is such a producer.
struct SweepLevel { const double *fac, *inv; const double* xsep; double* x; int n, m, P, flag0, up_flag0, up_stride, seg0, nseg, twisted, outer, store_left, x_last;
                    int merge, publish_g, skip_x, gflag0; double* gx; };
struct SweepArgs {
    LevelSrc root_src;
    LevelDst root_dst;
    int root_n;
    SweepLevel lv[ISLAM_PVGO_MAX_LEVELS];      // lv[0] = the level just below the root ... lv[nl-1] = the largest level
    int first_block[ISLAM_PVGO_MAX_LEVELS + 1];   // workgroup index where lv[i] starts (first_block[0] == 8)
    int nl;
    int* ready;               // per-segment words; ready[flag0 + p] == serial once segment p of that level is solved
    int serial;
    int root_twisted;         // the root is eliminated by both wavefronts of workgroup 0 (root_n <= BS_PAR_MAX)
    const double* outer_x;    // sharded solve: the solution at the rank's left cut node (9 doubles, a replicated level's x) ...
    int outer_flag;           // ... and the ready word of the segment that publishes it
    const double* fwd_src;    // sharded fused loop: the verdict block the decision kernel in front of this launch left in device memory ...
    double* fwd_dst;          // ... goes to the host's pinned slot from here (16 doubles, [15] = sequence number, last), whatever the gate says
};

#ifndef ISLAM_POLL_SLEEP
#define ISLAM_POLL_SLEEP 8          // s_sleep between two polls of a ready word (x 64 clocks); scripts/poll_sweep.sh
#endif
constexpr int READY_STRIDE = 32;      // ints between two ready words: one 128-byte line each (polled words spread over L2 channels)

__device__ __forceinline__ void wait_ready(const int* f, int serial, int* flags, int lane) {
    if (lane == 0) {
        int spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != serial) {
            __builtin_amdgcn_s_sleep(ISLAM_POLL_SLEEP);
            if (++spins > (1 << 22)) { atomicOr(flags, 2); break; }     // never observed; keeps a logic error from hanging the GPU
        }
    }
    asm volatile("" ::: "memory");      // what was published is read with ld_coherent AFTER this point: no acquire fence
}

__device__ __forceinline__ void publish_ready(int* f, int serial, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the wave's st_coherent stores have completed
    if (lane == 0) __hip_atomic_store(f, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 128 threads: the second wavefront only helps workgroup 0 with a twisted root elimination and exits everywhere else.
// (two waves per SIMD, i.e. at most 256 VGPRs: the ~1000 workgroups of the N = 5001 tree must all be resident -- at 268 VGPRs the second
// half of the level-0 segments started only when the first had finished: 17 -> 25 us)
__global__ __launch_bounds__(128, 2) void bt_downsweep_kernel(SweepArgs a, int* flags, Gate gate) {
    __shared__ __attribute__((aligned(16))) double lds[LDS_TWISTED > LDS_INFLUENCE ? LDS_TWISTED : LDS_INFLUENCE];
    if (a.fwd_src && blockIdx.x == gridDim.x - 1 && threadIdx.x < 64) {
        // (a one-workgroup decision kernel that waits for its own stores to host memory is 2 us longer -- on the critical path of
        // every trial; here the round trip over PCIe hides behind the sweep, in a workgroup that starts by waiting anyway)
        if (threadIdx.x < 15) __hip_atomic_store(&a.fwd_dst[threadIdx.x], a.fwd_src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_store(&a.fwd_dst[15], a.fwd_src[15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (gate_closed(gate)) return;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x;
    // A root of exactly three nodes (the N = 5001 tree) never leaves the workgroup: its two sweeps run in the staged (HELP) form --
    // no factor stores, no accumulations (no outer separators) -- and the back-substitution takes its factor rows straight from
    // the three LDS stages (forward sweep: nodes 0, 1 in its two stages; reverse sweep: node 2).  Root published 8.8 -> 7.x us
    // into the launch (scripts/probe_sweep.py).
    const bool root3 = a.root_twisted && a.root_n == 3;
    if (threadIdx.x >= 64) {
        if (b == 0 && root3) sweep_with_helper<0>(a.root_src, a.root_dst, 3, 3, 0, flags, 1, lane, lds, Gate{nullptr, 0.0}, 2);
        else if (b == 0 && a.root_twisted) eliminate_twisted(a.root_src, a.root_dst, a.root_n, a.root_n, 0, flags, 1, lane, lds);
        return;
    }
    if (b == 0) {                                   // root: eliminate + solve
        PROBE_WALL(lane == 0, 300);
        if (root3) sweep_with_helper<0>(a.root_src, a.root_dst, 3, 3, 0, flags, 0, lane, lds, Gate{nullptr, 0.0}, 2);
        else if (a.root_twisted) eliminate_twisted(a.root_src, a.root_dst, a.root_n, a.root_n, 0, flags, 0, lane, lds);
        else eliminate_segment(a.root_src, a.root_dst, a.root_n, a.root_n, 0, flags, lane, lds);
        PROBE_WALL(lane == 0, 301);
        double xn[9], xL[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) { xn[q] = 0.0; xL[q] = 0.0; }
        if (root3) {
            lds_sync();
            const int t = min(lane / 9, 2), r = lane - (lane / 9) * 9;          // lane 9t + r: row r of node t (lanes >= 27 duplicate node 2)
            const double* st = t == 0 ? lds : t == 1 ? lds + H_STAGE : lds + H_SWEEP;
            FacRow row;
#pragma unroll
            for (int i = 0; i < 9; ++i) { row.lt[i] = st[i * XS + r]; row.u[i] = st[(9 + i) * XS + r]; row.f[i] = st[(18 + i) * XS + r]; }
            row.y = st[27 * XS + r];
            row.iv = st[H_FST + H_XB + r];
            backsub_par_run_tw(a.root_dst.x, 0, 3, lane, xn, xL, row);
        } else if (a.root_twisted) {
            FacRow row;
            backsub_par_load(a.root_dst.fac, a.root_dst.inv, 0, a.root_n, lane, row);
            backsub_par_run_tw(a.root_dst.x, 0, a.root_n, lane, xn, xL, row);
        } else if (a.root_n <= BS_PAR_MAX) {
            FacRow row;
            backsub_par_load(a.root_dst.fac, a.root_dst.inv, 0, a.root_n, lane, row);
            backsub_par_run(a.root_dst.x, 0, a.root_n, lane, xn, xL, row);
        } else {
            backsub_segment(a.root_dst.fac, a.root_dst.inv, a.root_dst.x, 0, a.root_n, lane, xn, xL);
        }
        PROBE_WALL(lane == 0, 302);
        publish_ready(a.ready, a.serial, lane);     // ready[0] = the root
        PROBE_WALL(lane == 0, 303);
        return;
    }
    if (b < a.first_block[0]) return;               // padding so that every level starts at a multiple of 8 (XCD mapping)
    int li = 0;
    while (li + 1 < a.nl && b >= a.first_block[li + 1]) ++li;
    const SweepLevel L = a.lv[li];
    const int pw = xcd_index(b - a.first_block[li], L.nseg);   // blocks of a level are padded to a multiple of 8
    if (pw < 0) return;
    const int p = L.seg0 + pw;
    const bool outer_left = L.outer && pw == 0;
    const int stride = L.m + 1;
    const int c0 = p * stride;
    const int cnt = min(L.m, L.n - c0);
    const bool has_left = p > 0;
    const int sR = c0 + L.m;
    const bool has_right = sR < L.n;
    [[maybe_unused]] const bool pr = lane == 0 && (p == 1 || p == L.P / 2 || p == L.P - 1);     // probe build only
    [[maybe_unused]] const int po = (p == 1 ? 0 : p == L.P / 2 ? 50 : 100);
    PROBE_WALL(pr, po + 310 + 10 * li);
    const bool par = L.m <= BS_PAR_MAX;
    FacRow cur;
    if (par) backsub_par_load(L.fac, L.inv, c0, cnt, lane, cur);
    else load_facrow(L.fac, L.inv, c0 + cnt - 1, lane < 9 ? lane : 8, cur);
    // address-translation warm-up: touch the pages this wave will read (separators) and write (its part of x) while it
    // has nothing else to do; with hundreds of waves starting at once the page walks otherwise land on the critical path
    // (indices clamped to the arrays: x has n*9 entries, xsep (n/(m+1))*9; a read past the end of the caller's dx tensor
    // can fall off the end of a mapped allocation)
    const int nsep9 = (L.n / (L.m + 1)) * 9;
    double warm0 = L.xsep[min((has_left ? p - 1 : p) * 9 + (lane & 7), max(nsep9 - 1, 0))];
    double warm1 = L.x[min(c0 * 9 + lane, L.x_last)];
    __builtin_amdgcn_sched_barrier(0);
    // twisted segments: everything of the back-substitution that does not need the separators, now (influence_tw)
    // (only when the whole grid is resident, ~8 workgroups per CU: on longer chains the later workgroups do not wait, so the
    // extra 5 us would sit on their critical path -- N = 50 001: 254 vs 248 us per LM iteration)
    const bool infl = L.twisted && par && cnt >= 1 && gridDim.x <= 2048;
    double g[19];
    if (infl) influence_tw(cur, cnt, lane, lds, g);
    PROBE_WALL(pr, po + 318 + 10 * li);
    if (infl && L.publish_g) {                      // hand the influence matrices to the level below (row r of node t: 19 doubles)
        // straight from influence_tw's LDS copy, 512 contiguous bytes per store instruction (a lane storing its own row -- 19
        // stores, 152 bytes apart between lanes -- took 2.2 us: 855 separate write-through transactions)
        {
            const double* Gl = lds + 9 * BS_PAR_MAX * INF_FR;
            double* gp = L.gx + (size_t)c0 * 171;
            for (int e = lane; e < 171 * cnt; e += 64) st_coherent(gp + e, Gl[(e / 19) * INF_GS + (e % 19)]);
        }
        publish_ready(a.ready + (size_t)(L.gflag0 + p) * READY_STRIDE, a.serial, lane);
        PROBE_WALL(pr, po + 317 + 10 * li);                  // (probe build: G published)
    }
    if (infl && L.publish_g && L.skip_x) return;     // nobody reads this level's own solution: its nodes are separators of the level below
    if (infl && L.merge) {
        // Two levels in one hand-off.  This segment's separators q0 = p-1, q1 = p are nodes of the producer level; each is either
        // an interior node of a producer segment s -- x_q = G_q [1; -U(s-1); -U(s)], U = the solution one level further up -- or
        // a separator of the producer level, i.e. itself the node U(s).  Lanes 0-8 / 9-17 fetch row r of G_q0 / G_q1 while
        // everybody waits; when U arrives they evaluate x_q0, x_q1 (27 broadcasts, 18 FMAs), the wave broadcasts those (18 more)
        // and every lane takes its own dot product as usual.
        const SweepLevel P = a.lv[li - 1];
        const int ps = P.m + 1, nup = P.n / ps;
        const int q0 = p - 1, q1 = p;
        const int base = has_left ? q0 / ps : 0;
        const int s1 = has_right ? q1 / ps : base;
        const bool int0 = has_left && (q0 - base * ps) < P.m;
        const bool int1 = has_right && (q1 - s1 * ps) < P.m;
        const bool sh = s1 != base;                          // q1's producer segment is the next one: its U's are base, base+1
        // ALL ready words this segment depends on are polled at once, one word per lane: lanes 0 / 1 the G-ready words of the (up to
        // two) producer segments, lanes 2-4 the words of the (up to three) nodes of the level above the producer.  The rows of G are
        // requested the moment their words are seen -- whichever side arrives first no longer delays the other by a round trip
        // (the top pair sees the root's solution long before its producers' G, the bottom pair the other way round).
        const int rq = lane < 9 ? lane : lane - 9;           // row of G_q0 (lanes 0-8) / G_q1 (lanes 9-17)
        const bool mineint = lane < 9 ? int0 : (lane < 18 && int1);
        const bool minesep = lane < 9 ? (has_left && !int0) : (lane < 18 && has_right && !int1);
        const int* myf = nullptr;
        if (lane == 0 && int0) myf = a.ready + (size_t)(P.gflag0 + base) * READY_STRIDE;
        if (lane == 1 && int1 && (!int0 || sh)) myf = a.ready + (size_t)(P.gflag0 + s1) * READY_STRIDE;
        if (lane >= 2 && lane < 5) {
            const int j = base - 1 + (lane - 2);
            const bool dup = lane > 2 && j - 1 >= 0 && (j - 1) / P.up_stride == j / P.up_stride;      // same word as the lane before
            if (j >= 0 && j < nup && !dup) myf = a.ready + (size_t)(P.up_flag0 + j / P.up_stride) * READY_STRIDE;
        }
        double tq[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) tq[i] = 0.0;
        bool g_issued = false;
        for (int spins = 0;; ++spins) {
            const int ok = myf ? (__hip_atomic_load(myf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.serial) : 1;
            const unsigned long long seen = __ballot(ok);
            const bool gok = (seen & 3ull) == 3ull, xok = (seen & 0x1cull) == 0x1cull;
            if (gok && !g_issued) {
                PROBE_WALL(pr, po + 315 + 10 * li);          // (probe build: G-ready words seen)
#pragma unroll
                for (int i = 0; i < 6; ++i) {                // both matrices, contiguous: six loads per lane, all in flight
                    const int e = min(lane + 64 * i, 341);
                    const bool first = e < 171;
                    const double* gsrc = P.gx + (size_t)(first ? q0 : q1) * 171 + (first ? e : e - 171);
                    tq[i] = (first ? int0 : int1) ? ld_coherent(gsrc) : 0.0;
                }
                g_issued = true;
            }
            if (gok && xok) break;
            __builtin_amdgcn_s_sleep(ISLAM_POLL_SLEEP);
            if (spins > (1 << 22)) { if (lane == 0) atomicOr(flags, 2); break; }     // never observed; keeps a logic error from hanging the GPU
        }
        asm volatile("" ::: "memory");
        PROBE_WALL(pr, po + 311 + 10 * li);
        double uv = 0.0;
        {
            const int j = base - 1 + lane / 9;
            if (lane < 27 && j >= 0 && j < nup) uv = ld_coherent(&P.xsep[(size_t)(base - 1) * 9 + lane]);
        }
        double gq[19];
        {
            // the matrices go through LDS, then every lane picks its row
            double* Gq = lds;                                // [2][171]
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (lane + 64 * i < 342) Gq[lane + 64 * i] = tq[i];
            lds_sync();
            const double* gp = Gq + (lane < 9 ? 0 : 171) + rq * 19;
#pragma unroll
            for (int k = 0; k < 19; ++k) gq[k] = mineint ? gp[k] : 0.0;
            if (minesep) {                                   // the separator IS the node U(s): x_q[r] = -(-1) U(s)[r]
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    if (k == rq) gq[10 + k] = -1.0;
            }
        }
        PROBE_WALL(pr, po + 316 + 10 * li);                  // (probe build: rows of G in registers)
        const bool second = lane >= 9 && sh;                 // lanes of q1 when its segment is base+1: (UL, UR) = slots (1, 2)
        double xq = gq[0];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double u0 = bcast(uv, k), u1 = bcast(uv, 9 + k), u2 = bcast(uv, 18 + k);
            xq = fma(-gq[1 + k], second ? u1 : u0, xq);
            xq = fma(-gq[10 + k], second ? u2 : u1, xq);
        }
        double v = g[0];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            v = fma(-g[1 + k], bcast(xq, k), v);
            v = fma(-g[10 + k], bcast(xq, 9 + k), v);
        }
        PROBE_WALL(pr, po + 312 + 10 * li);
        if (lane < 9 * cnt) st_coherent(&L.x[(size_t)c0 * 9 + lane], v);
        if (has_right && lane >= 9 && lane < 18) st_coherent(&L.x[(size_t)sR * 9 + lane - 9], xq);
        PROBE_WALL(pr, po + 313 + 10 * li);
        if (li + 1 < a.nl) publish_ready(a.ready + (size_t)(L.flag0 + p) * READY_STRIDE, a.serial, lane);
        PROBE_WALL(pr, po + 314 + 10 * li);
        return;
    }
    // the solution cannot arrive before the root is solved and li levels above are expanded: stay off the memory system
    // until then (s_sleep 48 = 3072 clocks per level of distance -- deliberately short of the measured arrival times; the
    // influence matrices take ~5.5 us, about as long as the root: only the levels further down sleep on top of that)
    if (infl) { for (int i = 1; i <= li; ++i) __builtin_amdgcn_s_sleep(24); }
    else { for (int i = 0; i <= li; ++i) __builtin_amdgcn_s_sleep(48); }
    asm volatile("" ::"v"(warm0), "v"(warm1));
    // separators p-1 and p are nodes of the level above; node q there is published by its segment q / up_stride
    PROBE_WALL(pr, po + 315 + 10 * li);
    if (has_left)
        wait_ready(a.ready + (size_t)(outer_left ? a.outer_flag : L.up_flag0 + (p - 1) / L.up_stride) * READY_STRIDE, a.serial, flags, lane);
    PROBE_WALL(pr, po + 316 + 10 * li);
    if (has_right && (!has_left || outer_left || p / L.up_stride != (p - 1) / L.up_stride))
        wait_ready(a.ready + (size_t)(L.up_flag0 + p / L.up_stride) * READY_STRIDE, a.serial, flags, lane);
    PROBE_WALL(pr, po + 311 + 10 * li);
    // one load per lane (lanes 0-8: left separator, 9-17: right separator), then lane broadcasts
    double sv = 0.0;
    if (lane < 9 ? has_left : (lane < 18 && has_right))
        sv = ld_coherent((outer_left && lane < 9) ? a.outer_x + lane : &L.xsep[(size_t)(p - 1) * 9 + lane]);
    // the rank's left cut node has no segment of its own in the window: its row of this level's x (at level 0: the step the
    // trial needs for the rank's first link) is written by the window's first segment
    if (L.store_left && pw == 0 && lane < 9) st_coherent(&L.x[(size_t)(c0 - 1) * 9 + lane], sv);
#ifdef ISLAM_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PROBE_WALL(pr, po + 317 + 10 * li);
#endif
    double xn[9], xL[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        xL[q] = bcast(sv, q);
        xn[q] = bcast(sv, 9 + q);
    }
    if (has_right && lane >= 9 && lane < 18) st_coherent(&L.x[(size_t)sR * 9 + lane - 9], sv);
    PROBE_WALL(pr, po + 312 + 10 * li);
    if (infl) {                                      // x = G [1; -xL; -xR]: one dot product per lane (xn holds the right separator)
        double v = g[0];
#pragma unroll
        for (int q = 0; q < 9; ++q) v = fma(-g[1 + q], xL[q], v);
#pragma unroll
        for (int q = 0; q < 9; ++q) v = fma(-g[10 + q], xn[q], v);
        if (lane < 9 * cnt) st_coherent(&L.x[(size_t)c0 * 9 + lane], v);
    }
    else if (L.twisted) backsub_par_run_tw(L.x, c0, cnt, lane, xn, xL, cur);      // (twisted levels always have m <= BS_PAR_MAX)
    else if (par) backsub_par_run(L.x, c0, cnt, lane, xn, xL, cur);
    else backsub_run(L.fac, L.inv, L.x, c0, cnt, lane, xn, xL, cur);
    PROBE_WALL(pr, po + 313 + 10 * li);
    if (li + 1 < a.nl) publish_ready(a.ready + (size_t)(L.flag0 + p) * READY_STRIDE, a.serial, lane);
    PROBE_WALL(pr, po + 314 + 10 * li);
}

// ------------------------------------------------------------------------------------------
// (device state / report layout, TRParams, scheduler_step and lm_control: pvgo_internal.h)

// trial step: retract on a copy, new residuals, loss and trust-region denominator partials; the last block to
// finish sums the partials in index order (deterministic) and takes the LM decision (no separate control launch).
__global__ __launch_bounds__(64) void trial_kernel(const double* __restrict__ nodes, const double* __restrict__ vels,
                                                    const double* __restrict__ dx, const double* __restrict__ poses,
                                                    const double* __restrict__ drots, const double* __restrict__ dtrans,
                                                    const double* __restrict__ dvels, const double* __restrict__ dts,
                                                    const double* __restrict__ lin, int M, double* __restrict__ nodes_t,
                                                    double* __restrict__ vels_t, double* part, double* st, int* flags,
                                                    unsigned* ticket, TRParams tr, double* report, double seq,
                                                    const double* __restrict__ red_lin, const double* __restrict__ red_trial,
                                                    ReprojDev rp, int lin_stride, Gate gate, int* eflag2 = nullptr) {
    if (gate_closed(gate)) return;
    const int nblk = (M + 63) / 64;
    const int blk = xcd_index(blockIdx.x, nblk);
    int k = blk * 64 + threadIdx.x;
    double sq = 0.0, qd = 0.0;
    [[maybe_unused]] const bool pr = threadIdx.x == 0 && blk == 1;     // probe build only
    PROBE_AT(pr, 200);
    if (blk >= 0 && k < M) {
        const double* di = dx + (size_t)k * 9;
        const double* dj = dx + (size_t)(k + 1) * 9;
        V3<double> dri = ld3(di), dpi = ld3(di + 3), dvi = ld3(di + 6);
        V3<double> drj = ld3(dj), dpj = ld3(dj + 3), dvj = ld3(dj + 6);
        SE3<double> Xi = se3_mul(se3_exp(dri, dpi), se3_load(nodes + 7 * k));             // LieTensor.add_
        SE3<double> Xj = se3_mul(se3_exp(drj, dpj), se3_load(nodes + 7 * (k + 1)));
        V3<double> vi = ld3(vels + 3 * k) + dvi, vj = ld3(vels + 3 * (k + 1)) + dvj;
        double dt = dts[k];
        PROBE_AT(pr, 201);
        LinkRes r = link_residuals(Xi, Xj, vi, vj, se3_load(poses + 7 * k), ld4(drots + 4 * k), ld3(dtrans + 3 * k),
                                   ld3(dvels + 3 * k), dt);
        PROBE_AT(pr, 202);
        sq = dot(r.erho, r.erho) + dot(r.ephi, r.ephi) + dot(r.rv, r.rv) + dot(r.er, r.er) + dot(r.rt, r.rt);
        se3_store(Xi, nodes_t + 7 * k);
        vels_t[3 * k] = vi.x; vels_t[3 * k + 1] = vi.y; vels_t[3 * k + 2] = vi.z;
        if (k == M - 1) {
            se3_store(Xj, nodes_t + 7 * (k + 1));
            vels_t[3 * k + 3] = vj.x; vels_t[3 * k + 4] = vj.y; vels_t[3 * k + 5] = vj.z;
        }
        // -(J D)^T (2 R + J D) with the UNWEIGHTED J, R of the linearisation point (ppost.TrustRegion.update)
        double rec[LIN_C];
#pragma unroll
        for (int c = 0; c < LIN_C; ++c) rec[c] = lin[(size_t)c * lin_stride + k];
        M3<double> G = m3_load(rec + 6), C = m3_load(rec + 15), B = m3_load(rec + 27);
        V3<double> ddr = drj - dri, ddp = dpj - dpi;
        V3<double> j0 = G * ddr + C * ddp, j1 = G * ddp, j2 = dvi - dvj, j3 = B * ddp, j4 = ddr - dt * dvi;
        V3<double> R0{rec[0], rec[1], rec[2]}, R1{rec[3], rec[4], rec[5]}, R2{rec[36], rec[37], rec[38]},
            R3{rec[24], rec[25], rec[26]}, R4{rec[39], rec[40], rec[41]};
        qd = dot(j0, 2.0 * R0 + j0) + dot(j1, 2.0 * R1 + j1) + dot(j2, 2.0 * R2 + j2) + dot(j3, 2.0 * R3 + j3) +
             dot(j4, 2.0 * R4 + j4);
        if (red_lin) {           // reprojection rows: (J D)^T (2 R + J D) = u^T (2 b + S u), u = Ad(C^-1 X_i^-1)(d_j - d_i)
            sq += red_trial[(size_t)k * RP_REC + 27];
            double u[RP_NSUM];
#pragma unroll
            for (int i = 0; i < RP_NSUM; ++i) u[i] = red_lin[(size_t)k * RP_REC + i];
            M3<double> Ra, Ta;
            reproj_adjoint(rp, se3_load(nodes + 7 * k), Ra, Ta);
            const V3<double> ua = Ra * ddr + Ta * ddp, ub = Ra * ddp;
            const V3<double> sa = sym_from(u, 0, 0) * ua + sym_from(u, 0, 3) * ub;
            const V3<double> sb = tmul(sym_from(u, 0, 3), ua) + sym_from(u, 3, 3) * ub;
            const V3<double> ba{u[21], u[22], u[23]}, bb{u[24], u[25], u[26]};
            qd += dot(ua, 2.0 * ba + sa) + dot(ub, 2.0 * bb + sb);
        }
    }
    PROBE_AT(pr, 203);
    sq = wave_sum(sq);
    qd = wave_sum(qd);
    // (write-through stores + completion wait instead of a release fence -- an agent-scope fence walks the XCD's whole L2)
    if (threadIdx.x == 0 && blk >= 0) { st_coherent(&part[2 * blk], sq); st_coherent(&part[2 * blk + 1], qd); }
    PROBE_AT(pr, 204);
    if (st == nullptr) return;           // stage-level call: no control
    // ---- last block takes the decision
    int last_block = 0;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this block's partial sums are written through
        last_block = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    }
    last_block = __builtin_amdgcn_readfirstlane(last_block);
    PROBE_AT(pr, 205);
    if (!last_block) return;
    PROBE_AT(threadIdx.x == 0, 206);
    double s = 0.0, q = 0.0;                                          // (agent-coherent loads of the other blocks' partial sums)
    for (int i = threadIdx.x; i < nblk; i += 64) {
        s += __hip_atomic_load(&part[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q += __hip_atomic_load(&part[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (threadIdx.x == 0) {
        *ticket = 0u;
        bool failed = flags[0] != 0;
        flags[0] = 0;
        if (eflag2) { failed = failed || *eflag2 != 0; *eflag2 = 0; }     // (level 0 of the solve ran inside trial_elim_kernel)
        lm_control(s, q, st, failed, tr, report, seq);
    }
    PROBE_AT(threadIdx.x == 0, 207);
}

// The LM loop's trial step and the NEXT step's linearisation in one launch (the trial point is the next linearisation
// point whenever the trial is accepted -- the common case; after a reject the output buffer is simply overwritten).
// Wave 0: one lane per link (lane 0 = halo link shared with the previous workgroup): retraction, residuals at the trial
// point, partial sum of the loss.  Wave 1, concurrently: the trust-region term (J D)^T (2R + J D) of the same links (it
// needs the step and the OLD linearisation only); after the workgroup barrier its lane 0 publishes both partial sums and
// bumps the ticket without waiting for it.  The decision is taken by one extra workgroup that polls the ticket -- on
// nobody's critical path -- so the host learns the verdict while the linearisation is still being written.
// Then as linbuild_kernel: Jacobians, weighted pieces, node blocks, coalesced copy.
__global__ __launch_bounds__(LB_THREADS) void trial_lin_kernel(
    const double* __restrict__ nodes, const double* __restrict__ vels, const double* __restrict__ dx,
    const double* __restrict__ poses, const double* __restrict__ drots, const double* __restrict__ dtrans,
    const double* __restrict__ dvels, const double* __restrict__ dts, const double* __restrict__ lin, int N,
    double* __restrict__ nodes_t, double* __restrict__ vels_t, double* part, double* st, int* flags, unsigned* ticket,
    TRParams tr, double* report, double seq, const double* __restrict__ red_lin, const double* __restrict__ red_trial,
    ReprojDev rp, LinWeights W, double* __restrict__ lin_o, double* __restrict__ Hd_o, double* __restrict__ Ho_o,
    double* __restrict__ rhs_o, Gate gate, int* eflag2 = nullptr) {
    __shared__ double sl[64][LB_REC];
    __shared__ double s_sq;
    extern __shared__ __attribute__((aligned(16))) double lb_out[];
    const int M = N - 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nlb = (N + LB_NODES - 1) / LB_NODES;
    if (blockIdx.x == gridDim.x - 1) {
        // The deciding workgroup (one extra workgroup behind the grid, dispatched last): waits until every workgroup has
        // published its partial sums (ticket == nlb), adds them in index order (deterministic) and takes the LM decision --
        // on nobody's critical path: the other workgroups go straight on to the next linearisation.
        if (wave != 0 || gate_closed(gate)) return;
        PROBE_WALL(lane == 0, 420);
        __builtin_amdgcn_s_sleep(64);                  // the sums cannot be there before the residuals are evaluated (~2 us)
        if (lane == 0) {
            int spins = 0;
            while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)nlb) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 22)) { atomicOr(flags, 2); break; }     // never observed; a logic error must not hang the GPU
            }
        }
        asm volatile("" ::: "memory");                 // the partial sums are read with ld_coherent after this point
        double ssum = 0.0, qsum = 0.0;
        for (int i = lane; i < nlb; i += 64) {
            ssum += ld_coherent(&part[2 * i]);
            qsum += ld_coherent(&part[2 * i + 1]);
        }
        ssum = wave_sum(ssum);
        qsum = wave_sum(qsum);
        if (lane == 0) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool failed = flags[0] != 0;
            flags[0] = 0;
            if (eflag2) { failed = failed || *eflag2 != 0; *eflag2 = 0; }     // (level 0 of the solve ran inside trial_elim_kernel)
            lm_control(ssum, qsum, st, failed, tr, report, seq);
        }
        PROBE_WALL(lane == 0, 421);
        return;
    }
    const int blk = xcd_index(blockIdx.x, nlb);
    if (blk < 0 || gate_closed(gate)) return;
    const int L = blk * LB_NODES - 1 + lane;
    const bool valid = L >= 0 && L < M && lane <= LB_NODES;      // lanes past the block's last link idle when LB_NODES < 63
    const bool owns = valid && (lane > 0 || blk == 0);
#ifdef ISLAM_PROBE
    const bool pr = threadIdx.x == 0 && blk == nlb / 2;
#endif
    PROBE_WALL(pr, 400);
    SE3<double> Xi{}, Xj{};
    V3<double> vi{}, vj{};
    LinkRes r{};
    double dt = 0.0;
    double qd = 0.0;
    if (wave == 0) {                                   // the trial point and its residuals
        double sq = 0.0;
        if (valid) {
            const double* di = dx + (size_t)L * 9;
            const V3<double> dri = ld3(di), dpi = ld3(di + 3), dvi = ld3(di + 6);
            const V3<double> drj = ld3(di + 9), dpj = ld3(di + 12), dvj = ld3(di + 15);
            Xi = se3_mul(se3_exp(dri, dpi), se3_load(nodes + 7 * L));                      // LieTensor.add_
            Xj = se3_mul(se3_exp(drj, dpj), se3_load(nodes + 7 * (L + 1)));
            vi = ld3(vels + 3 * L) + dvi;
            vj = ld3(vels + 3 * (L + 1)) + dvj;
            dt = dts[L];
            PROBE_WALL(pr, 401);
            r = link_residuals(Xi, Xj, vi, vj, se3_load(poses + 7 * L), ld4(drots + 4 * L), ld3(dtrans + 3 * L),
                               ld3(dvels + 3 * L), dt);
            PROBE_WALL(pr, 402);
            if (owns) {
                sq = dot(r.erho, r.erho) + dot(r.ephi, r.ephi) + dot(r.rv, r.rv) + dot(r.er, r.er) + dot(r.rt, r.rt);
                if (red_lin) sq += red_trial[(size_t)L * RP_REC + 27];
            }
        }
        PROBE_WALL(pr, 403);
        sq = wave_sum(sq);
        if (lane == 0) s_sq = sq;
    } else if (wave == 1) {
        // concurrently: -(J D)^T (2 R + J D) with the UNWEIGHTED J, R of the linearisation point (ppost.TrustRegion.update);
        // it needs the step and the old linearisation only, not the trial residuals
        if (owns) {
            const double* di = dx + (size_t)L * 9;
            const V3<double> dri = ld3(di), dpi = ld3(di + 3), dvi = ld3(di + 6);
            const V3<double> drj = ld3(di + 9), dpj = ld3(di + 12), dvj = ld3(di + 15);
            const double dtl = dts[L];
            double rec[LIN_C];
#pragma unroll
            for (int c = 0; c < LIN_C; ++c) rec[c] = lin[(size_t)c * M + L];
            const M3<double> G = m3_load(rec + 6), C = m3_load(rec + 15), B = m3_load(rec + 27);
            const V3<double> ddr = drj - dri, ddp = dpj - dpi;
            const V3<double> j0 = G * ddr + C * ddp, j1 = G * ddp, j2 = dvi - dvj, j3 = B * ddp, j4 = ddr - dtl * dvi;
            const V3<double> R0{rec[0], rec[1], rec[2]}, R1{rec[3], rec[4], rec[5]}, R2{rec[36], rec[37], rec[38]},
                R3{rec[24], rec[25], rec[26]}, R4{rec[39], rec[40], rec[41]};
            qd = dot(j0, 2.0 * R0 + j0) + dot(j1, 2.0 * R1 + j1) + dot(j2, 2.0 * R2 + j2) + dot(j3, 2.0 * R3 + j3) +
                 dot(j4, 2.0 * R4 + j4);
            if (red_lin) {       // reprojection rows: u^T (2 b + S u), u = Ad(C^-1 X_i^-1)(d_j - d_i)
                double u[RP_NSUM];
#pragma unroll
                for (int i = 0; i < RP_NSUM; ++i) u[i] = red_lin[(size_t)L * RP_REC + i];
                M3<double> Ra, Ta;
                reproj_adjoint(rp, se3_load(nodes + 7 * L), Ra, Ta);
                const V3<double> ua = Ra * ddr + Ta * ddp, ub = Ra * ddp;
                const V3<double> sa = sym_from(u, 0, 0) * ua + sym_from(u, 0, 3) * ub;
                const V3<double> sb = tmul(sym_from(u, 0, 3), ua) + sym_from(u, 3, 3) * ub;
                const V3<double> ba{u[21], u[22], u[23]}, bb{u[24], u[25], u[26]};
                qd += dot(ua, 2.0 * ba + sa) + dot(ub, 2.0 * bb + sb);
            }
        }
        qd = wave_sum(qd);
    }
    PROBE_WALL(pr, 404);
    __syncthreads();
    if (wave == 1 && lane == 0) {
        // publish: write-through stores + completion wait instead of a release fence (an agent-scope release walks the
        // L2), then the ticket -- fire and forget, nobody in this workgroup waits for it
        st_coherent(&part[2 * blk], s_sq);
        st_coherent(&part[2 * blk + 1], qd);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PROBE_WALL(pr, 405);
    if (wave == 0 && valid && owns) {
        se3_store(Xi, nodes_t + 7 * L);
        vels_t[3 * L] = vi.x; vels_t[3 * L + 1] = vi.y; vels_t[3 * L + 2] = vi.z;
        if (L == M - 1) {
            se3_store(Xj, nodes_t + 7 * (L + 1));
            vels_t[3 * L + 3] = vj.x; vels_t[3 * L + 4] = vj.y; vels_t[3 * L + 5] = vj.z;
        }
    }
    if (lin_o == nullptr) return;            // trial only (the last trial of a run: nothing is linearised at its trial point)
    if (wave == 0 && valid) {
        PROBE_WALL(pr, 406);
        M3<double> G, C, B;
        link_jacobians(r, G, C, B);
        PROBE_WALL(pr, 407);
        link_emit(r, G, C, B, dt, L, M, owns, W, lin_o, sl[lane], red_trial, rp, Xi);
        PROBE_WALL(pr, 408);
    }
    __syncthreads();
    PROBE_WALL(pr, 409);
    nodes_build_copy(sl, lb_out, blk, N, W, Hd_o, Ho_o, rhs_o);
    PROBE_WALL(pr, 410);
}

// ------------------------------------------------------------------------------------------
// Small graphs: the WHOLE LM loop of run_pvgo in ONE launch of ONE workgroup.
// The reference optimises a window of batch_size + 1 = 9 nodes per training step (train.py:253-263, run_kitti.sh:8): one
// block-tridiagonal segment (bt_top_kernel with a single level) and one block of links (trial_lin_kernel).  Launched per stage that is
// two dependent launches and one host round trip per LM trial -- ~40 us per trial on an idle GPU, and 3-4x that inside the bilevel
// step, where every one of those launches waits for a CU slot beside the frozen nets' convolution kernels of the next batch
// (scripts/vio_chain.py: the PVGO stage took 0.7 ms alone and 3.0 ms in the pipelined step).  Here the host launches once and polls
// once: the loop of islam_pvgo_run_chain's launch-per-stage branch -- damped solve, trial step, TrustRegion.update, accept / reject,
// StopOnPlateau, the linearisation at an accepted trial point, the re-linearisation after a failed solve -- runs on the device with
// the same device functions in the same order (same numbers: tests/test_pvgo_gpu.py compares both loops), workgroup barriers where
// the launch-per-stage loop has kernel boundaries.  The linearisation at the trial point is built only once the trial is accepted.
struct SmallArgs {
    double *nodes, *vels;                               // the iterate (in / out)
    const double *poses, *drots, *dtrans, *dvels, *dts;
    int N;
    double *nodes_t, *vels_t, *dx;
    double *LIN[2], *HD[2], *HO[2], *RH[2];              // linearisation buffers; [0] holds the linearisation of the initial iterate
    double* loss_part;
    double* st;
    int* flags;
    TRParams tr;
    LinWeights W;
    LevelDst dst;                                       // factor storage of the single level
    double* report;                                     // pinned host block: [0] loss [2] damping [10] status [11] trials [13] steps [15] marker
    double* trace;                                      // pinned host rows (trial loss, damping, accepted) or nullptr
    int trace_cap;
    double marker;
};

__global__ __launch_bounds__(LB_THREADS) void small_lm_kernel(SmallArgs a) {
    __shared__ double sl[64][LB_REC];
    __shared__ double s_sq;
    __shared__ int s_verdict;
    extern __shared__ __attribute__((aligned(16))) double small_dyn[];      // lb_out of nodes_build_copy | the solve's column copies
    double* lb_out = small_dyn;
    double* lds_solve = small_dyn + LB_DYN_BYTES / (int)sizeof(double);
    const int N = a.N, M = N - 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int L = lane - 1;                                      // link of this lane (block 0 of linbuild / trial_lin)
    const bool valid = L >= 0 && L < M && lane <= LB_NODES;
    const ReprojDev rp{};
    int pb = 0, trials = 0, status = ISLAM_OK;
    double *cur_n = a.nodes, *cur_v = a.vels, *tri_n = a.nodes_t, *tri_v = a.vels_t;
    for (;;) {
        // ---- damped solve on buffer pb (bt_top_kernel with one level: the diagonal is damped in place, cumulatively over retries)
        if (wave == 0) {
            LevelSrc src{};
            src.level0 = 1; src.Hd = a.HD[pb]; src.Ho = a.HO[pb]; src.rhs0 = a.RH[pb]; src.state = a.st; src.damping_override = 0.0;
            eliminate_segment(src, a.dst, N, N, 0, a.flags, lane, lds_solve);
            double xn[9], xL[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) { xn[q] = 0.0; xL[q] = 0.0; }
            backsub_segment(a.dst.fac, a.dst.inv, a.dx, 0, N, lane, xn, xL);
        }
        __syncthreads();                                         // dx is visible to the workgroup
        // ---- the trial point, its residuals (wave 0) and the trust-region term of the old linearisation (wave 1): trial_lin_kernel
        SE3<double> Xi{}, Xj{};
        V3<double> vi{}, vj{};
        LinkRes r{};
        double dt = 0.0, qd = 0.0;
        if (wave == 0) {
            double sq = 0.0;
            if (valid) {
                const double* di = a.dx + (size_t)L * 9;
                const V3<double> dri = ld3(di), dpi = ld3(di + 3), dvi = ld3(di + 6);
                const V3<double> drj = ld3(di + 9), dpj = ld3(di + 12), dvj = ld3(di + 15);
                Xi = se3_mul(se3_exp(dri, dpi), se3_load(cur_n + 7 * L));                      // LieTensor.add_
                Xj = se3_mul(se3_exp(drj, dpj), se3_load(cur_n + 7 * (L + 1)));
                vi = ld3(cur_v + 3 * L) + dvi;
                vj = ld3(cur_v + 3 * (L + 1)) + dvj;
                dt = a.dts[L];
                r = link_residuals(Xi, Xj, vi, vj, se3_load(a.poses + 7 * L), ld4(a.drots + 4 * L), ld3(a.dtrans + 3 * L),
                                   ld3(a.dvels + 3 * L), dt);
                sq = dot(r.erho, r.erho) + dot(r.ephi, r.ephi) + dot(r.rv, r.rv) + dot(r.er, r.er) + dot(r.rt, r.rt);
            }
            sq = wave_sum(sq);
            if (lane == 0) s_sq = sq;
        } else if (wave == 1) {
            if (valid) {
                const double* lin = a.LIN[pb];
                const double* di = a.dx + (size_t)L * 9;
                const V3<double> dri = ld3(di), dpi = ld3(di + 3), dvi = ld3(di + 6);
                const V3<double> drj = ld3(di + 9), dpj = ld3(di + 12), dvj = ld3(di + 15);
                const double dtl = a.dts[L];
                double rec[LIN_C];
#pragma unroll
                for (int c = 0; c < LIN_C; ++c) rec[c] = lin[(size_t)c * M + L];
                const M3<double> G = m3_load(rec + 6), C = m3_load(rec + 15), B = m3_load(rec + 27);
                const V3<double> ddr = drj - dri, ddp = dpj - dpi;
                const V3<double> j0 = G * ddr + C * ddp, j1 = G * ddp, j2 = dvi - dvj, j3 = B * ddp, j4 = ddr - dtl * dvi;
                const V3<double> R0{rec[0], rec[1], rec[2]}, R1{rec[3], rec[4], rec[5]}, R2{rec[36], rec[37], rec[38]},
                    R3{rec[24], rec[25], rec[26]}, R4{rec[39], rec[40], rec[41]};
                qd = dot(j0, 2.0 * R0 + j0) + dot(j1, 2.0 * R1 + j1) + dot(j2, 2.0 * R2 + j2) + dot(j3, 2.0 * R3 + j3) +
                     dot(j4, 2.0 * R4 + j4);
            }
            qd = wave_sum(qd);
        }
        __syncthreads();
        if (wave == 0 && valid) {                                // the trial iterate
            se3_store(Xi, tri_n + 7 * L);
            tri_v[3 * L] = vi.x; tri_v[3 * L + 1] = vi.y; tri_v[3 * L + 2] = vi.z;
            if (L == M - 1) {
                se3_store(Xj, tri_n + 7 * (L + 1));
                tri_v[3 * L + 3] = vj.x; tri_v[3 * L + 4] = vj.y; tri_v[3 * L + 5] = vj.z;
            }
        }
        if (wave == 1 && lane == 0) {                            // the LM decision (one lane, as in the deciding workgroup of trial_lin_kernel)
            const bool failed = a.flags[0] != 0;
            a.flags[0] = 0;
            const int v = lm_control(s_sq, qd, a.st, failed, a.tr, nullptr, (double)(trials + 1));
            if (a.trace && trials < a.trace_cap && v < 3) {
                __hip_atomic_store(&a.trace[3 * trials], a.st[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&a.trace[3 * trials + 1], a.st[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&a.trace[3 * trials + 2], v == 1 ? 0.0 : 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            s_verdict = v;
        }
        __syncthreads();
        const int verdict = s_verdict;
        ++trials;
        if (verdict == 0) {
            // accepted, continue: the linearisation at the trial point into the other buffer (trial_lin_kernel's second half), then swap
            if (wave == 0 && valid) {
                M3<double> G, C, B;
                link_jacobians(r, G, C, B);
                link_emit(r, G, C, B, dt, L, M, true, a.W, a.LIN[1 - pb], sl[lane], nullptr, rp, Xi);
            }
            __syncthreads();
            nodes_build_copy(sl, lb_out, 0, N, a.W, a.HD[1 - pb], a.HO[1 - pb], a.RH[1 - pb]);
            __syncthreads();
            pb = 1 - pb;
            double* t;
            t = cur_n; cur_n = tri_n; tri_n = t;
            t = cur_v; cur_v = tri_v; tri_v = t;
            continue;
        }
        if (verdict == 1) continue;                              // rejected: same iterate, same (cumulatively damped) linearisation
        if (verdict == 2) {                                      // accepted, StopOnPlateau says stop
            double* t;
            t = cur_n; cur_n = tri_n; tri_n = t;
            t = cur_v; cur_v = tri_v; tri_v = t;
            break;
        }
        status = ISLAM_ENOTPD;                                   // "Linear solver failed. Breaking optimization step..."
        if (verdict == 4) break;
        // PyPose keeps looping through the scheduler: same iterate, new linearisation (linbuild_kernel's body)
        if (wave == 0) {
            double sq = 0.0;
            if (valid) {
                const SE3<double> Yi = se3_load(cur_n + 7 * L), Yj = se3_load(cur_n + 7 * (L + 1));
                const double dtl = a.dts[L];
                const LinkRes rr = link_residuals(Yi, Yj, ld3(cur_v + 3 * L), ld3(cur_v + 3 * (L + 1)), se3_load(a.poses + 7 * L),
                                                  ld4(a.drots + 4 * L), ld3(a.dtrans + 3 * L), ld3(a.dvels + 3 * L), dtl);
                M3<double> G, C, B;
                link_jacobians(rr, G, C, B);
                sq = dot(rr.erho, rr.erho) + dot(rr.ephi, rr.ephi) + dot(rr.rv, rr.rv) + dot(rr.er, rr.er) + dot(rr.rt, rr.rt);
                link_emit(rr, G, C, B, dtl, L, M, true, a.W, a.LIN[pb], sl[lane], nullptr, rp, Yi);
            }
            sq = wave_sum(sq);
            if (lane == 0) a.loss_part[0] = sq;
        }
        __syncthreads();
        nodes_build_copy(sl, lb_out, 0, N, a.W, a.HD[pb], a.HO[pb], a.RH[pb]);
        __syncthreads();
    }
    // ---- the result goes back into the caller's arrays; one record for the host
    __syncthreads();
    if (cur_n != a.nodes) {
        for (int e = threadIdx.x; e < 7 * N; e += LB_THREADS) a.nodes[e] = cur_n[e];
        for (int e = threadIdx.x; e < 3 * N; e += LB_THREADS) a.vels[e] = cur_v[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&a.report[0], a.st[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.report[2], a.st[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.report[10], (double)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.report[11], (double)trials, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.report[13], a.st[12], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // the copy above has left the CU before the host is told (it may launch readers next)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&a.report[15], a.marker, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------
// The LM loop's steady state in ONE launch: trial step + loss / trust-region partial sums of trial t, the linearisation at the
// trial point, AND the level-0 elimination of the next solve straight out of LDS.
//
// The elimination of step t+1 damps its diagonal with TrustRegion.update's output for trial t, which needs sums over ALL links --
// a grid-wide dependency between the linearisation and the first pivot.  It is broken by SPECULATION: the next damping has three
// possible values (radius x up / kept / x down); every workgroup assumes that the trial is accepted and that TrustRegion.update takes
// the branch it took for the previous trial (speculated_damping: an LM run stays in one regime for many trials -- on the
// 5000-frame bench graph the radius is kept on all ten).  The deciding workgroup (one extra workgroup, as in trial_lin_kernel)
// validates the guess when the sums are in and bumps the run-ahead epoch otherwise (verdict 5, or any of the non-"accepted,
// continue" verdicts): the launches queued behind this one (upper levels, down-sweep) turn into no-ops and the host redoes the
// solve on the launched level-0 kernel from the linearisation this kernel wrote to global memory (undamped: LevelSrc::hist).
//
// Workgroup = FZ_S = 4 CONSECUTIVE level-0 segments = one contiguous stretch of G = 4 (m+1) nodes, eight wavefronts, one workgroup
// per CU (the first version gave every segment its own three-wave workgroup: 834 wavefronts each ran the whole SE(3) arithmetic of
// its 7 links on 7 of 64 lanes, two or three of them per SIMD -- 20 us before the first pivot, scripts/probe_fused.py).  The
// per-link arithmetic is cut along its natural seams so that no wavefront carries a long instruction stream:
//   A  wave 0, one lane per NODE: retraction X <- Exp(dx) X (LieTensor.add_), the trial iterate goes to LDS and to global memory
//   B  wave 0, one lane per LINK: pose-graph residual Log(P^-1 Xi^-1 Xj), its Jacobian blocks G, C and their weighted products
//      wave 1, one lane per link: IMU rotation / velocity / translation residuals, B and its products   (concurrently)
//      wave 2, one lane per link: trust-region term (J D)^T (2R + J D) from the OLD linearisation        (concurrently)
//      wave 3: sums both partial sums over the workgroup's links and publishes them
//   C  all waves: node blocks Hd / Ho / rhs of the stretch in LDS, one 3x3 sub-block per thread (type-major: a wave builds one or
//      two kinds of block, no divergence); the same blocks go to global memory, lane-contiguous (fallback solves read them)
//   D  waves 2s, 2s+1: twisted elimination of segment s, columns read from the LDS blocks (eliminate_twisted)
// The sums of products are formed in the order of link_emit / nodes_build_copy, so the linearisation is bit-identical to
// linbuild_kernel's.  Saves per LM iteration: one launch, the 13.7 MB round trip of Hd / Ho / rhs through HBM on the critical
// path, and the level-0 kernel's first dependent loads.
constexpr int FZ_S = 4;                                    // segments per workgroup
constexpr int FZ_HELPERS = FZ_S / 2;                       // helper waves, two segments each
constexpr int FZ_THREADS = (2 * FZ_S + FZ_HELPERS) * 64;
constexpr int FZ_MAXM = BS_PAR_MAX;
constexpr int FZ_G = FZ_S * (FZ_MAXM + 1);                 // nodes of a workgroup's stretch (at most)
constexpr int FZ_XT = 10;                                  // retracted node: t 3 | q 4 | v 3
constexpr int FZ_RV = 35;                                  // pose-graph pieces of a link: Srr 9 | Srp 9 | Spp 9 | gr 3 | gp 3 | e.e 1 | pad
constexpr int FZ_RI = 25;                                  // IMU pieces: Spp 9 | w3 rt 3 | gp 3 | rv 3 | rt 3 | dt 1 | rv.rv, er.er, rt.rt
constexpr int fz_even(int x) { return (x + 1) & ~1; }
constexpr int FZ_OFF_XT = 0;
constexpr int FZ_OFF_SV = fz_even(FZ_OFF_XT + (FZ_G + 2) * FZ_XT);
constexpr int FZ_OFF_SI = fz_even(FZ_OFF_SV + (FZ_G + 1) * FZ_RV);
constexpr int FZ_OFF_SUM = fz_even(FZ_OFF_SI + (FZ_G + 1) * FZ_RI);
constexpr int FZ_OFF_HD = FZ_OFF_SUM + 4;
constexpr int FZ_OFF_HO = fz_even(FZ_OFF_HD + FZ_G * 81);
constexpr int FZ_OFF_RHS = fz_even(FZ_OFF_HO + (FZ_G + 1) * 81);
constexpr int FZ_OFF_TW = fz_even(FZ_OFF_RHS + FZ_G * 9);
constexpr int FZ_LDS = FZ_OFF_TW + FZ_S * LDS_TW4;
constexpr int FZ_LDS_BYTES = FZ_LDS * (int)sizeof(double);

struct FusedArgs {
    const double *nodes, *vels, *dx, *poses, *drots, *dtrans, *dvels, *dts, *lin;    // iterate, step, measurements, OLD linearisation
    int N;
    double *nodes_t, *vels_t;                     // trial iterate
    double* part;
    double* st;
    int* flags;
    unsigned* ticket;
    TRParams tr;
    double* report;
    double seq;
    LinWeights W;
    double *lin_o, *Hd_o, *Ho_o, *rhs_o;          // linearisation at the trial point (diagonal clamped, UNDAMPED)
    LevelDst dst;                                 // level-0 factor and products
    int m, P, nwg;                                // level-0 segment length / count, workgroups (each takes <= FZ_S consecutive segments)
    int* eflag;                                   // solver-error word of THIS elimination
    int* eflag_prev;                              // ... of the level-0 elimination of the solve whose trial is evaluated here
    const double* loss_part0;                     // first trial of a run only: the partial sums of the initial loss (linbuild_kernel) --
    int nlb0;                                     // the deciding wave does control_begin_kernel's job on the way (one launch less per run)
    // ---- one rank of the sharded loop (run_chain_sharded_fused; all zero on a single GPU except Ms = N - 1)
    int Ms;                                       // row stride of lin / lin_o (links of the WHOLE chain)
    int shard;                                    // 1: no deciding workgroup -- the sums leave in part[0 .. 2 nwg), part[2 nwg] = failed-pivot
                                                  //    word of the solve whose trial this is; the decision follows the all-reduce
    int seg_lo;                                   // first level-0 segment of the rank; P = the rank's segment count; N = one past the rank's
                                                  //    right outer separator (the link beyond it belongs to the next rank)
    int own_left;                                 // the rank has a left outer separator: its first workgroup owns the link that leaves it,
    double* share;                                //    steps it, and writes that link's UNCLAMPED, undamped part of its block + rhs here (90)
    int open_right;                               // node N - 1 is shared with the next rank: its diagonal is clamped after the all-reduce
    int trial_only;                               // the last optimizer step of a sharded run: only the trial iterate and the sums (no blocks, no elimination)
};

__device__ __forceinline__ M3<double> m3_zero() { return M3<double>{0, 0, 0, 0, 0, 0, 0, 0, 0}; }
__device__ __forceinline__ void put33(double* h, const M3<double>& b) {        // 3x3 block into a row-major 9-wide matrix
    h[0] = b.a00; h[1] = b.a01; h[2] = b.a02; h[9] = b.a10; h[10] = b.a11; h[11] = b.a12; h[18] = b.a20; h[19] = b.a21; h[20] = b.a22;
}

__global__ __launch_bounds__(FZ_THREADS, 3) void trial_elim_kernel(FusedArgs a, Gate gate) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* xt = lds + FZ_OFF_XT;
    double* sv = lds + FZ_OFF_SV;
    double* si = lds + FZ_OFF_SI;
    double* s_sum = lds + FZ_OFF_SUM;
    double* Hd_l = lds + FZ_OFF_HD;
    double* Ho_l = lds + FZ_OFF_HO;
    double* rhs_l = lds + FZ_OFF_RHS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int N = a.N, M = N - 1;
    const size_t Ms = (size_t)a.Ms;
    if (!a.shard && blockIdx.x == gridDim.x - 1) {
        // the deciding workgroup (see trial_lin_kernel): sums the partials in index order, LM decision, validates the speculation
        if (wave != 0 || gate_closed(gate)) return;
        const bool first = a.dx == nullptr;
        const double d_spec = (first || a.trial_only) ? -1.0 : speculated_damping(a.st, a.tr);      // (trial only: no solve is running ahead)
        __builtin_amdgcn_s_sleep(64);
        if (lane == 0) {
            int spins = 0;
            while (__hip_atomic_load(a.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)a.nwg) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 22)) {
                    // Never observed; a logic error must not hang the GPU.  The partial sums are incomplete, so this is NOT a trial
                    // outcome: close the run-ahead gate (everything queued behind becomes a no-op), leave the ticket alone (stragglers
                    // of this launch still count into it; control_init_kernel clears it for the next run) and hand the host verdict 9,
                    // which fails the run with ISLAM_EHIP.
                    atomicOr(a.flags, 8);
                    a.st[14] = -1.0;
                    if (a.report) {
                        __hip_atomic_store(&a.report[12], 9.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_store(&a.report[15], a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    return;
                }
            }
        }
        asm volatile("" ::: "memory");
        double ssum = 0.0, qsum = 0.0;
        for (int i = lane; i < a.nwg; i += 64) {
            ssum += ld_coherent(&a.part[2 * i]);
            qsum += ld_coherent(&a.part[2 * i + 1]);
        }
        ssum = wave_sum(ssum);
        qsum = wave_sum(qsum);
        double l0 = 0.0;
        if (a.loss_part0) {                       // self.loss of the very first optimizer.step(): summed like control_begin_kernel does
            for (int i = lane; i < a.nlb0; i += 64) l0 += a.loss_part0[i];
            l0 = wave_sum(l0);
        }
        if (lane == 0) {
            __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (first) {                          // the linearisation of the initial iterate: its loss is the run's first `last`
                a.st[0] = ssum; a.st[1] = ssum; a.st[8] = 0.0; a.st[11] = 1.0; a.st[12] = 0.0; a.st[13] = 0.0;
                return;
            }
            const bool failed = a.flags[0] != 0 || *a.eflag_prev != 0;
            a.flags[0] = 0;
            *a.eflag_prev = 0;
            if (a.loss_part0) { a.st[0] = l0; a.st[1] = l0; a.st[8] = 0.0; a.st[11] = 1.0; a.st[12] = 0.0; a.st[13] = 0.0; }
            lm_control(ssum, qsum, a.st, failed, a.tr, a.report, a.seq, d_spec);
        }
        return;
    }
    const int wg = xcd_index(blockIdx.x, a.nwg);
    if (wg < 0 || gate_closed(gate)) return;
    // first: the run's first linearisation (no step yet: dx == nullptr) -- the iterate itself instead of a trial point, no
    // trust-region term, no decision, and the elimination uses the initial damping (nothing to speculate on)
    const bool first = a.dx == nullptr;
    const double d_spec = first ? a.st[2] : speculated_damping(a.st, a.tr);       // (read before this workgroup publishes: the decision comes later)
    // the level's segments are dealt out evenly: workgroup wg takes segments [wg P / nwg, (wg+1) P / nwg) -- three or four of them
    // on the 5000-frame graph, so that all 256 CUs share the level's pivots
    const int seg0 = a.seg_lo + (int)(((long long)wg * a.P) / a.nwg), seg1 = a.seg_lo + (int)(((long long)(wg + 1) * a.P) / a.nwg);
    const int m = a.m, stride = m + 1, G = (seg1 - seg0) * stride;
    const bool ownl = a.own_left && wg == 0;                     // (sharded: the link cb-1 -> cb has no other owner on this rank)
    const int cb = seg0 * stride;                                // first node of the stretch; links cb-1 .. cb+G-1, nodes cb-1 .. cb+G
    [[maybe_unused]] const bool fpr = lane == 0 && (wg == 1 || wg == a.nwg / 2);     // probe build only
    [[maybe_unused]] const int fpo = 600 + (wg == 1 ? 0 : 100) + 12 * wave;
    PROBE_WALL(fpr, fpo);
    // (probe build: entry / exit of every workgroup's wave 0 -- launch ramp and drain of the grid, scripts/probe_fused.py)
    PROBE_WALL(threadIdx.x == 0 && blockIdx.x < 300, 300 + blockIdx.x);
    if (threadIdx.x == 0) s_sum[2] = 0.0;                        // LevelSrc::zero of the elimination below (same address space as the blocks)
    // ---- A: retraction, one lane per node
    if (wave == 0) {
        const int k = cb - 1 + lane;
        if (lane < G + 2 && k >= 0 && k < N) {
            SE3<double> X = se3_load(a.nodes + 7 * k);
            V3<double> v = ld3(a.vels + 3 * k);
            if (!first) {
                const double* d = a.dx + (size_t)k * 9;
                X = se3_mul(se3_exp(ld3(d), ld3(d + 3)), X);                                              // LieTensor.add_
                v = v + ld3(d + 6);
            }
            double* o = xt + lane * FZ_XT;
            se3_store(X, o);
            o[7] = v.x; o[8] = v.y; o[9] = v.z;
            if ((lane >= 1 || ownl) && lane <= G) {                  // the stretch's own nodes
                se3_store(X, a.nodes_t + 7 * k);
                a.vels_t[3 * k] = v.x; a.vels_t[3 * k + 1] = v.y; a.vels_t[3 * k + 2] = v.z;
            }
        }
    }
    lds_barrier();
    PROBE_WALL(fpr, fpo + 1);
    // ---- B: one lane per link j (link L = cb-1+j joins the nodes in xt[j], xt[j+1])
    {
        const int L = cb - 1 + lane;
        const bool valid = lane <= G && L >= 0 && L < M;
        const bool owns = valid && (lane >= 1 || ownl);              // links cb .. cb+G-1 belong to this stretch
        const double* xi = xt + lane * FZ_XT;
        const double* xj = xi + FZ_XT;
        if (wave == 0 && valid) {
            const SE3<double> Xi = se3_load(xi), Xj = se3_load(xj);
            const SE3<double> pre = se3_mul(se3_inv(se3_load(a.poses + 7 * L)), se3_inv(Xi));
            V3<double> erho, ephi;
            se3_log(se3_mul(pre, Xj), erho, ephi);
            const M3<double> Ji = so3_Jl_inv(ephi);
            const M3<double> R = qmat(pre.q);
            const M3<double> Gm = Ji * R;
            const M3<double> C = Ji * (skew(pre.t) * R - se3_Q(erho, ephi) * Gm);
            if (owns) {
                double* lo = a.lin_o + L;
                lo[0] = erho.x; lo[Ms] = erho.y; lo[2 * Ms] = erho.z;
                lo[3 * Ms] = ephi.x; lo[4 * Ms] = ephi.y; lo[5 * Ms] = ephi.z;
                double rec[18];
                m3_store(Gm, rec);
                m3_store(C, rec + 9);
#pragma unroll
                for (int c = 0; c < 18; ++c) lo[(size_t)(6 + c) * Ms] = rec[c];
            }
            const M3<double> Gt = transpose(Gm), Ct = transpose(C);
            const M3<double> GtG = Gt * Gm;
            double* o = sv + lane * FZ_RV;
            m3_store(a.W.w0 * GtG + a.W.w3 * m3_identity<double>(), o);
            m3_store(a.W.w0 * (Gt * C), o + 9);
            m3_store(a.W.w0 * (Ct * C + GtG), o + 18);
            const V3<double> gr = a.W.w0 * (Gt * erho), gp = a.W.w0 * (Ct * erho + Gt * ephi);
            o[27] = gr.x; o[28] = gr.y; o[29] = gr.z; o[30] = gp.x; o[31] = gp.y; o[32] = gp.z;
            o[33] = dot(erho, erho) + dot(ephi, ephi);
        } else if (wave == 1 && valid) {
            const V3<double> ti = ld3(xi), tj = ld3(xj), vi = ld3(xi + 7), vj = ld3(xj + 7);
            const Q4<double> qi = ld4(xi + 3), qj = ld4(xj + 3);
            const double dt = a.dts[L];
            const V3<double> rv = ld3(a.dvels + 3 * L) - (vj - vi);
            const Q4<double> rpre = qmul(qinv(ld4(a.drots + 4 * L)), qinv(qi));
            const V3<double> er = so3_log(qmul(rpre, qj));
            const V3<double> rt = (tj - ti) - (dt * vi + ld3(a.dtrans + 3 * L));
            const M3<double> B = so3_Jl_inv(er) * qmat(rpre);
            if (owns) {
                double* lo = a.lin_o + L;
                lo[24 * Ms] = er.x; lo[25 * Ms] = er.y; lo[26 * Ms] = er.z;
                double rec[9];
                m3_store(B, rec);
#pragma unroll
                for (int c = 0; c < 9; ++c) lo[(size_t)(27 + c) * Ms] = rec[c];
                lo[36 * Ms] = rv.x; lo[37 * Ms] = rv.y; lo[38 * Ms] = rv.z;
                lo[39 * Ms] = rt.x; lo[40 * Ms] = rt.y; lo[41 * Ms] = rt.z;
            }
            const M3<double> Bt = transpose(B);
            double* o = si + lane * FZ_RI;
            m3_store(a.W.w2 * (Bt * B), o);
            const V3<double> w3rt = a.W.w3 * rt, gp = a.W.w2 * (Bt * er);
            o[9] = w3rt.x; o[10] = w3rt.y; o[11] = w3rt.z; o[12] = gp.x; o[13] = gp.y; o[14] = gp.z;
            o[15] = rv.x; o[16] = rv.y; o[17] = rv.z; o[18] = rt.x; o[19] = rt.y; o[20] = rt.z; o[21] = dt;
            o[22] = dot(rv, rv); o[23] = dot(er, er); o[24] = dot(rt, rt);
        } else if (wave == 2) {
            // -(J D)^T (2 R + J D) with the UNWEIGHTED J, R of the linearisation point (ppost.TrustRegion.update)
            double qd = 0.0;
            if (owns && !first) {
                const double* di = a.dx + (size_t)L * 9;
                const V3<double> dri = ld3(di), dpi = ld3(di + 3), dvi = ld3(di + 6);
                const V3<double> drj = ld3(di + 9), dpj = ld3(di + 12), dvj = ld3(di + 15);
                const double dtl = a.dts[L];
                double rec[LIN_C];
#pragma unroll
                for (int c = 0; c < LIN_C; ++c) rec[c] = a.lin[(size_t)c * Ms + L];
                const M3<double> Gm = m3_load(rec + 6), C = m3_load(rec + 15), B = m3_load(rec + 27);
                const V3<double> ddr = drj - dri, ddp = dpj - dpi;
                const V3<double> j0 = Gm * ddr + C * ddp, j1 = Gm * ddp, j2 = dvi - dvj, j3 = B * ddp, j4 = ddr - dtl * dvi;
                const V3<double> R0{rec[0], rec[1], rec[2]}, R1{rec[3], rec[4], rec[5]}, R2{rec[36], rec[37], rec[38]},
                    R3{rec[24], rec[25], rec[26]}, R4{rec[39], rec[40], rec[41]};
                qd = dot(j0, 2.0 * R0 + j0) + dot(j1, 2.0 * R1 + j1) + dot(j2, 2.0 * R2 + j2) + dot(j3, 2.0 * R3 + j3) +
                     dot(j4, 2.0 * R4 + j4);
            }
            qd = wave_sum(qd);
            if (lane == 0) s_sum[1] = qd;
        }
    }
    lds_barrier();
    PROBE_WALL(fpr, fpo + 2);
    if (a.trial_only) {
        // (sharded loop: nothing follows an accepted trial of the last optimizer step -- the trial iterate is stored, the sums go to the
        // all-reduce; no node blocks, no elimination)
        if (wave == 2 * FZ_S + FZ_HELPERS - 1) {
            const int L = cb - 1 + lane;
            double sq = 0.0;
            if ((lane >= 1 || ownl) && lane <= G && L < M) {
                const double* o = si + lane * FZ_RI;
                sq = sv[lane * FZ_RV + 33] + o[22] + o[23] + o[24];
            }
            sq = wave_sum(sq);
            if (lane == 0 && a.shard) {
                a.part[2 * wg] = sq;
                a.part[2 * wg + 1] = s_sum[1];
                if (wg == 0) {
                    a.part[2 * a.nwg] = (!first && (a.flags[0] != 0 || *a.eflag_prev != 0)) ? 1.0 : 0.0;
                    a.flags[0] = 0;
                    *a.eflag_prev = 0;
                }
            } else if (lane == 0) {                  // (single GPU: the deciding workgroup of this launch takes the decision)
                st_coherent(&a.part[2 * wg], sq);
                st_coherent(&a.part[2 * wg + 1], s_sum[1]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    // (the two partial sums are published after the block build, by a helper wave: a wave that waits for its write-through stores
    // here arrives ~1 us late at the barrier behind the build and holds the whole workgroup -- measured: blocks 1.4 -> 0.6 us)
    // ---- C: node blocks.  Node i of the stretch (k = cb+i) takes link slots i (k-1) and i+1 (k); coupling i (k' = cb-1+i) is link slot i.
    // One 3x3 sub-block per thread, 32 slots per kind of block: a half-wave builds ONE kind (9 of Hd, 9 of Ho, the right-hand side)
    {
        const double w1 = a.W.w1, w3 = a.W.w3;
        const M3<double> I = m3_identity<double>();
        for (int it = threadIdx.x; it < 19 * 32; it += FZ_THREADS) {
            const int ty = it >> 5, i = it & 31;
            if (ty < 9) {
                const int br = ty / 3, bc = ty - br * 3, k = cb + i;
                if (i >= G || k >= N) continue;
                const bool hp = k > 0, hn = k < M;
                const double* v0 = sv + i * FZ_RV;
                const double* v1 = v0 + FZ_RV;
                const double* i0 = si + i * FZ_RI;
                const double* i1 = i0 + FZ_RI;
                M3<double> blk = m3_zero();
                if (br == 0 && bc == 0) { if (hp) blk = blk + m3_load(v0); if (hn) blk = blk + m3_load(v1); }
                else if (br + bc == 1) {
                    if (hp) blk = blk + m3_load(v0 + 9);
                    if (hn) blk = blk + m3_load(v1 + 9);
                    if (br == 1) blk = transpose(blk);
                } else if (br == 1 && bc == 1) {
                    if (hp) blk = blk + (m3_load(v0 + 18) + m3_load(i0));
                    if (hn) blk = blk + (m3_load(v1 + 18) + m3_load(i1));
                } else if (br == 2 && bc == 2) {
                    double hvv = 0.0;
                    if (hp) hvv += w1;
                    if (hn) { const double d = i1[21]; hvv += w1 + w3 * d * d; }
                    blk = hvv * I;
                } else if (br + bc == 2) { blk = (hn ? w3 * i1[21] : 0.0) * I; }
                if (br == bc && !(a.open_right && k == N - 1)) {         // A.diagonal().clamp_(min, max)
                    blk.a00 = fmin(fmax(blk.a00, a.W.vmin), a.W.vmax);
                    blk.a11 = fmin(fmax(blk.a11, a.W.vmin), a.W.vmax);
                    blk.a22 = fmin(fmax(blk.a22, a.W.vmin), a.W.vmax);
                }
                put33(Hd_l + i * 81 + br * 27 + bc * 3, blk);
            } else if (ty < 18) {
                const int b = ty - 9, br = b / 3, bc = b - br * 3, k = cb - 1 + i;    // coupling k -> k+1
                if (i > G) continue;
                M3<double> blk = m3_zero();
                if (k >= 0 && k < M) {
                    const double* v1 = sv + i * FZ_RV;
                    const double* i1 = si + i * FZ_RI;
                    if (br == 0 && bc == 0) blk = -1.0 * m3_load(v1);
                    else if (br == 0 && bc == 1) blk = -1.0 * m3_load(v1 + 9);
                    else if (br == 1 && bc == 0) blk = -1.0 * transpose(m3_load(v1 + 9));
                    else if (br == 1 && bc == 1) blk = -1.0 * (m3_load(v1 + 18) + m3_load(i1));
                    else if (br == 2 && bc == 0) blk = (-w3 * i1[21]) * I;
                    else if (br == 2 && bc == 2) blk = (-w1) * I;
                }
                put33(Ho_l + i * 81 + br * 27 + bc * 3, blk);
            } else {
                const int k = cb + i;
                if (i >= G || k >= N) continue;
                const double* v0 = sv + i * FZ_RV;
                const double* v1 = v0 + FZ_RV;
                const double* i0 = si + i * FZ_RI;
                const double* i1 = i0 + FZ_RI;
                V3<double> gr{0, 0, 0}, gp{0, 0, 0}, gv{0, 0, 0};
                if (k > 0) {
                    gr = gr + (ld3(v0 + 27) + ld3(i0 + 9)); gp = gp + (ld3(v0 + 30) + ld3(i0 + 12));
                    gv = gv - w1 * ld3(i0 + 15);
                }
                if (k < M) {
                    const double d = i1[21];
                    gr = gr - (ld3(v1 + 27) + ld3(i1 + 9)); gp = gp - (ld3(v1 + 30) + ld3(i1 + 12));
                    gv = gv + w1 * ld3(i1 + 15) - (w3 * d) * ld3(i1 + 18);
                }
                double* bb = rhs_l + i * 9;
                bb[0] = -gr.x; bb[1] = -gr.y; bb[2] = -gr.z; bb[3] = -gp.x; bb[4] = -gp.y; bb[5] = -gp.z;
                bb[6] = -gv.x; bb[7] = -gv.y; bb[8] = -gv.z;
            }
        }
        // sharded: what the link cb-1 -> cb adds to the block and the right-hand side of node cb-1, the PREVIOUS rank's right outer
        // separator (the `hn` terms above with link slot 0).  Unclamped and undamped: the block is a sum over two ranks
        // (shard_pack_kernel adds this part to the separator's row of the exchange buffer, shard_decide_kernel clamps the sum).
        if (ownl && threadIdx.x < 10) {
            const int ty = threadIdx.x;
            const double* v1 = sv;
            const double* i1 = si;
            const double d = i1[21];
            if (ty < 9) {
                const int br = ty / 3, bc = ty - br * 3;
                M3<double> blk = m3_zero();
                if (br == 0 && bc == 0) blk = blk + m3_load(v1);
                else if (br + bc == 1) { blk = blk + m3_load(v1 + 9); if (br == 1) blk = transpose(blk); }
                else if (br == 1 && bc == 1) blk = blk + (m3_load(v1 + 18) + m3_load(i1));
                else if (br == 2 && bc == 2) blk = (w1 + w3 * d * d) * I;
                else if (br + bc == 2) blk = (w3 * d) * I;
                put33(a.share + br * 27 + bc * 3, blk);
            } else {
                const V3<double> gr = V3<double>{0, 0, 0} - (ld3(v1 + 27) + ld3(i1 + 9)), gp = V3<double>{0, 0, 0} - (ld3(v1 + 30) + ld3(i1 + 12));
                const V3<double> gv = V3<double>{0, 0, 0} + w1 * ld3(i1 + 15) - (w3 * d) * ld3(i1 + 18);
                double* bb = a.share + 81;
                bb[0] = -gr.x; bb[1] = -gr.y; bb[2] = -gr.z; bb[3] = -gp.x; bb[4] = -gp.y; bb[5] = -gp.z;
                bb[6] = -gv.x; bb[7] = -gv.y; bb[8] = -gv.z;
            }
        }
    }
    lds_barrier();
    PROBE_WALL(fpr, fpo + 3);
    // ---- D: waves 2s, 2s+1 eliminate segment s of the stretch; helper wave 2 FZ_S + h serves segments 2h, 2h+1 (twisted_helper) after
    // it has copied its half of the linearisation to global memory.  Every wave executes the same number of barriers: the forward
    // step count of a full segment.
    LevelSrc src{};
    src.level0 = 1;
    src.Hd = Hd_l - (ptrdiff_t)cb * 81;
    src.Ho = Ho_l - (ptrdiff_t)(cb - 1) * 81;
    src.rhs0 = rhs_l - (ptrdiff_t)cb * 9;
    src.state = nullptr;
    src.damping_override = d_spec;
    src.hist = 1;
    src.zero = s_sum + 2;
    const int nbar = m >= 3 ? m / 2 + 1 : m;
    double* tw = lds + FZ_OFF_TW;
    PROBE_WALL(fpr, fpo + 4);
    if (wave < 2 * FZ_S) {
        // waves s and FZ_S + s sweep segment s forwards / backwards: a workgroup's waves go to the CU's four SIMDs round-robin, so
        // every SIMD gets one forward (three node steps) and one reverse sweep (two) instead of two of a kind
        const int seg = wave % FZ_S, p = seg0 + seg;
        if (p < seg1) sweep_with_helper<2>(src, a.dst, N, m, p, a.eflag, wave / FZ_S, lane, tw + seg * LDS_TW4, Gate{nullptr, 0.0}, nbar);
        else { for (int t = 0; t < nbar; ++t) lds_barrier(); }
    } else {
        if (wave == 2 * FZ_S + FZ_HELPERS - 1) {
            // unweighted loss of the stretch's own links (the terms in the order of link_residuals' sum), then publish both partial
            // sums: write-through stores + completion wait instead of a release fence, then the ticket (fire and forget)
            const int L = cb - 1 + lane;
            double sq = 0.0;
            if ((lane >= 1 || ownl) && lane <= G && L < M) {
                const double* o = si + lane * FZ_RI;
                sq = sv[lane * FZ_RV + 33] + o[22] + o[23] + o[24];
            }
            sq = wave_sum(sq);
            if (lane == 0 && a.shard) {              // the sums and the decision come after the all-reduce (shard_pack / shard_decide)
                a.part[2 * wg] = sq;
                a.part[2 * wg + 1] = s_sum[1];
                if (wg == 0) {
                    a.part[2 * a.nwg] = (!first && (a.flags[0] != 0 || *a.eflag_prev != 0)) ? 1.0 : 0.0;
                    a.flags[0] = 0;
                    *a.eflag_prev = 0;
                }
            } else if (lane == 0) {
                st_coherent(&a.part[2 * wg], sq);
                st_coherent(&a.part[2 * wg + 1], s_sum[1]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // the linearisation to global memory, lane-contiguous: the fallback solves (reject, mis-speculated damping, failed solve)
        // read it from there
        const int ht = threadIdx.x - 2 * FZ_S * 64, HT = FZ_HELPERS * 64;
        const int no = min(G, N - cb);                                    // nodes this stretch owns
        for (int e = ht; e < no * 81; e += HT) a.Hd_o[(size_t)cb * 81 + e] = Hd_l[e];
        for (int e = ht; e < no * 9; e += HT) a.rhs_o[(size_t)cb * 9 + e] = rhs_l[e];
        const int k0 = max(cb - 1, 0), k1 = min(cb + G - 2, M - 1);       // couplings this stretch owns (cb+G-1 is the next one's first)
        const double* hs = Ho_l + (k0 - (cb - 1)) * 81;
        for (int e = ht; e < (k1 - k0 + 1) * 81; e += HT) a.Ho_o[(size_t)k0 * 81 + e] = hs[e];
        const int hw = wave - 2 * FZ_S;
        HelpSeg sg[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int seg = 2 * hw + q, p = seg0 + seg;
            if (p < seg1) sg[q] = help_seg(seg_geom(N, m, p), p, tw + seg * LDS_TW4);
            else { sg[q] = HelpSeg{}; sg[q].on = false; }
        }
        twisted_helper<2>(a.dst, sg, nbar, lane);
    }
    PROBE_WALL(fpr, fpo + 5);
    PROBE_WALL(threadIdx.x == 0 && blockIdx.x < 300, 0 + blockIdx.x);
}

// state <- [damping = 1 / radius, radius, down, run-ahead epoch 1], everything else and the four flag words zero
// ready16 != nullptr: the down-sweep's ready words are zeroed by the same launch (n16 16-byte items over the whole grid) -- the separate
// fill launch in front of every run_pvgo cost ~2.5 us of the run
__global__ __launch_bounds__(256) void control_init_kernel(double* __restrict__ st, int* __restrict__ flags, double radius, double down,
                                                           uint4* __restrict__ ready16 = nullptr, unsigned n16 = 0) {
    const int t = threadIdx.x;
    if (blockIdx.x == 0) {
        if (t < STATE_DOUBLES) st[t] = (t == 2 || t == STATE_HIST) ? 1.0 / radius : t == 3 ? radius : t == 4 ? down : (t == 14 || t == 15) ? 1.0 : 0.0;     // [15]: first guess "radius kept"
        if (t < 8) flags[t] = 0;
    }
    for (unsigned i = blockIdx.x * 256u + t; i < n16; i += gridDim.x * 256u) ready16[i] = uint4{0u, 0u, 0u, 0u};
}

__global__ __launch_bounds__(64) void control_begin_kernel(const double* __restrict__ loss_part, int nblk,
                                                            double* __restrict__ st, int* flags) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) s += loss_part[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        st[0] = s;                                         // self.loss of the very first optimizer.step()
        st[1] = s;                                         // self.last = self.loss
        st[8] = 0.0;
        st[11] = 1.0;
        st[12] = 0.0;
        st[13] = 0.0;
        flags[0] = 0;
    }
}

__global__ __launch_bounds__(64) void retract_kernel(const double* __restrict__ nodes, const double* __restrict__ vels,
                                                      const double* __restrict__ dx, double sign, int N,
                                                      double* __restrict__ nodes_o, double* __restrict__ vels_o) {
    int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= N) return;
    const double* d = dx + (size_t)k * 9;
    SE3<double> X = se3_mul(se3_exp(sign * ld3(d), sign * ld3(d + 3)), se3_load(nodes + 7 * k));
    se3_store(X, nodes_o + 7 * k);
    V3<double> v = ld3(vels + 3 * k) + sign * ld3(d + 6);
    vels_o[3 * k] = v.x; vels_o[3 * k + 1] = v.y; vels_o[3 * k + 2] = v.z;
}

// VO factor of an ARBITRARY edge (i, j) (loop closures; pvgo.py:36-39): residual e = Log(P^-1 Xi^-1 Xj) and the blocks
// G, C of d e / d delta_j = [[G, C],[0, G]] (d e / d delta_i = -that).  out: (24, E) component-major.
__global__ __launch_bounds__(64) void vo_edge_linearize_kernel(const double* __restrict__ nodes, const int64_t* __restrict__ edges,
                                                                const double* __restrict__ poses, int E, double* __restrict__ out) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= E) return;
    SE3<double> Xi = se3_load(nodes + 7 * edges[2 * e]), Xj = se3_load(nodes + 7 * edges[2 * e + 1]);
    SE3<double> pre = se3_mul(se3_inv(se3_load(poses + 7 * e)), se3_inv(Xi));
    V3<double> rho, phi;
    se3_log(se3_mul(pre, Xj), rho, phi);
    M3<double> Ji = so3_Jl_inv(phi);
    M3<double> R = qmat(pre.q);
    M3<double> G = Ji * R;
    M3<double> C = Ji * (skew(pre.t) * R - se3_Q(rho, phi) * G);
    double rec[24];
    rec[0] = rho.x; rec[1] = rho.y; rec[2] = rho.z; rec[3] = phi.x; rec[4] = phi.y; rec[5] = phi.z;
    m3_store(G, rec + 6);
    m3_store(C, rec + 15);
#pragma unroll
    for (int c = 0; c < 24; ++c) out[(size_t)c * E + e] = rec[c];
}

// ---- general topology: dense A from block pieces (no dense J) ----
struct EdgeNormal { M3<double> Srr, Srp, Spp; V3<double> gr, gp; };

__device__ __forceinline__ EdgeNormal edge_normal(const double* __restrict__ vo, int E, int e) {
    double rec[24];
#pragma unroll
    for (int c = 0; c < 24; ++c) rec[c] = vo[(size_t)c * E + e];
    const V3<double> er{rec[0], rec[1], rec[2]}, ep{rec[3], rec[4], rec[5]};
    const M3<double> G = m3_load(rec + 6), C = m3_load(rec + 15);
    const M3<double> Gt = transpose(G), Ct = transpose(C);
    EdgeNormal o;
    o.Srr = Gt * G;
    o.Srp = Gt * C;
    o.Spp = Ct * C + o.Srr;
    o.gr = Gt * er;
    o.gp = Ct * er + Gt * ep;
    return o;
}

// one lane per node: diagonal block + right-hand side (fixed summation order over the node's edge ends), chain coupling
__global__ __launch_bounds__(64) void dense_nodes_kernel(const double* __restrict__ Hd, const double* __restrict__ Ho,
                                                          const double* __restrict__ rhs_chain, const double* __restrict__ vo,
                                                          const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ node_adj,
                                                          double w0, int N, int E, double* __restrict__ A, double* __restrict__ rhs) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= N) return;
    const size_t ld = (size_t)9 * N;
    const M3<double> Z{0, 0, 0, 0, 0, 0, 0, 0, 0};
    M3<double> Srr = Z, Srp = Z, Spp = Z;
    V3<double> gr{0, 0, 0}, gp{0, 0, 0};
    for (int64_t a = node_ptr[k]; a < node_ptr[k + 1]; ++a) {
        const int64_t code = node_adj[a];
        const EdgeNormal en = edge_normal(vo, E, (int)(code >> 1));
        Srr = Srr + en.Srr; Srp = Srp + en.Srp; Spp = Spp + en.Spp;
        if (code & 1) { gr = gr + en.gr; gp = gp + en.gp; } else { gr = gr - en.gr; gp = gp - en.gp; }
    }
    double blk[81];
#pragma unroll
    for (int i = 0; i < 81; ++i) blk[i] = Hd[(size_t)k * 81 + i];
    double add[36];
    m3_store(w0 * Srr, add); m3_store(w0 * Srp, add + 9); m3_store(w0 * Spp, add + 18); m3_store(w0 * transpose(Srp), add + 27);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            blk[r * 9 + c] += add[r * 3 + c];
            blk[r * 9 + 3 + c] += add[9 + r * 3 + c];
            blk[(3 + r) * 9 + c] += add[27 + r * 3 + c];
            blk[(3 + r) * 9 + 3 + c] += add[18 + r * 3 + c];
        }
    double* d = A + (size_t)9 * k * ld + 9 * k;
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int c = 0; c < 9; ++c) d[r * ld + c] = blk[r * 9 + c];
    const double* rc = rhs_chain + (size_t)k * 9;
    double* b = rhs + (size_t)k * 9;
    b[0] = rc[0] - w0 * gr.x; b[1] = rc[1] - w0 * gr.y; b[2] = rc[2] - w0 * gr.z;
    b[3] = rc[3] - w0 * gp.x; b[4] = rc[4] - w0 * gp.y; b[5] = rc[5] - w0 * gp.z;
    b[6] = rc[6]; b[7] = rc[7]; b[8] = rc[8];
    if (k < N - 1) {
        double* up = A + (size_t)9 * k * ld + 9 * (k + 1);
        double* lo = A + (size_t)9 * (k + 1) * ld + 9 * k;
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const double v = Ho[(size_t)k * 81 + r * 9 + c];
                up[r * ld + c] = v;
                lo[c * ld + r] = v;
            }
    }
}

// one lane per edge: the two off-diagonal blocks -w0 S (and its transpose) of an arbitrary edge (i, j)
__global__ __launch_bounds__(64) void dense_edges_kernel(const double* __restrict__ vo, const int64_t* __restrict__ edges, double w0,
                                                          int N, int E, double* __restrict__ A) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= E) return;
    const int64_t i = edges[2 * e], j = edges[2 * e + 1];
    if (i == j) return;
    const EdgeNormal en = edge_normal(vo, E, e);
    double S[36];       // row-major 6x6 [[Srr, Srp],[Srp^T, Spp]]
    const M3<double> Spr = transpose(en.Srp);
    double t[9];
    m3_store(en.Srr, t);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) S[r * 6 + c] = t[r * 3 + c];
    m3_store(en.Srp, t);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) S[r * 6 + 3 + c] = t[r * 3 + c];
    m3_store(Spr, t);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) S[(3 + r) * 6 + c] = t[r * 3 + c];
    m3_store(en.Spp, t);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) S[(3 + r) * 6 + 3 + c] = t[r * 3 + c];
    const size_t ld = (size_t)9 * N;
    double* ij = A + (size_t)9 * i * ld + 9 * j;
    double* ji = A + (size_t)9 * j * ld + 9 * i;
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            atomicAdd(&ij[r * ld + c], -w0 * S[r * 6 + c]);
            atomicAdd(&ji[c * ld + r], -w0 * S[r * 6 + c]);
        }
}

__global__ __launch_bounds__(64) void vo_loss_fwd_kernel(const double* __restrict__ nodes, const int64_t* __restrict__ edges,
                                                          const double* __restrict__ poses, int E, double* __restrict__ err6,
                                                          double* __restrict__ tl, double* __restrict__ rl) {
    int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= E) return;
    SE3<double> Xi = se3_load(nodes + 7 * edges[2 * e]), Xj = se3_load(nodes + 7 * edges[2 * e + 1]);
    SE3<double> P = se3_load(poses + 7 * e);
    V3<double> rho, phi;
    se3_log(se3_mul(se3_mul(se3_inv(P), se3_inv(Xi)), Xj), rho, phi);
    double* o = err6 + 6 * (size_t)e;
    o[0] = rho.x; o[1] = rho.y; o[2] = rho.z; o[3] = phi.x; o[4] = phi.y; o[5] = phi.z;
    tl[e] = dot(rho, rho);
    rl[e] = dot(phi, phi);
}

// PyPose autograd: g_E = g_e Jl^-1(e) ; g_P = -g_E Ad(P^-1), stored as a 7-vector with a trailing 0
__global__ __launch_bounds__(64) void vo_loss_bwd_kernel(const double* __restrict__ poses, const double* __restrict__ err6,
                                                          const double* __restrict__ g_trans, const double* __restrict__ g_rot,
                                                          int E, double* __restrict__ grad) {
    int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= E) return;
    const double* er = err6 + 6 * (size_t)e;
    V3<double> rho = ld3(er), phi = ld3(er + 3);
    V3<double> gr = (2.0 * g_trans[e]) * rho, gp = (2.0 * g_rot[e]) * phi;
    M3<double> Ji = so3_Jl_inv(phi);
    M3<double> Q = se3_Q(rho, phi);
    // row-vector times Jl^-1 = [[Ji, -Ji Q Ji],[0, Ji]]
    V3<double> a = tmul(Ji, gr);
    V3<double> b = tmul(Ji, gp) - tmul(Ji, tmul(Q, a));
    SE3<double> Pi = se3_inv(se3_load(poses + 7 * e));
    M3<double> R = qmat(Pi.q);
    // row-vector times Ad(Pi) = [[R, [t]x R],[0, R]]
    V3<double> o0 = tmul(R, a);
    V3<double> o1 = tmul(R, tmul(skew(Pi.t), a)) + tmul(R, b);
    double* g = grad + 7 * (size_t)e;
    g[0] = -o0.x; g[1] = -o0.y; g[2] = -o0.z; g[3] = -o1.x; g[4] = -o1.y; g[5] = -o1.z; g[6] = 0.0;
}

__global__ __launch_bounds__(64) void align_kernel(const double* __restrict__ nodes, const double* __restrict__ vels,
                                                    const double* __restrict__ target, int N, double* __restrict__ nodes_o,
                                                    double* __restrict__ vels_o) {
    int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= N) return;
    SE3<double> T = se3_load(target), S = se3_load(nodes);
    SE3<double> rel = se3_mul(T, se3_inv(S));
    Q4<double> rq = qmul(T.q, qinv(S.q));
    se3_store(se3_mul(rel, se3_load(nodes + 7 * k)), nodes_o + 7 * k);
    V3<double> v = qact(rq, ld3(vels + 3 * k));
    vels_o[3 * k] = v.x; vels_o[3 * k + 1] = v.y; vels_o[3 * k + 2] = v.z;
}

// ------------------------------------------------------------------------------------------
// host side
struct LevelPlan { int n, m, P, nsep; };
constexpr int MAXL = ISLAM_PVGO_MAX_LEVELS;
constexpr int TOPW = 1;                        // wavefronts of the top kernel's workgroup (see plan_levels)
struct SolvePlan { LevelPlan lv[MAXL]; int nl; int top; int twisted; };   // levels >= top run inside bt_top_kernel

// dependent node steps of one segment of m interior nodes
static inline int segment_steps(int m, bool twisted) { return (twisted && m >= 3) ? m / 2 + 1 : m; }

// Level tree.  The critical path is a chain of dependent node steps (~2.3 us each: eliminate + back-substitute) plus
// ~4 us per level boundary (launch + the first dependent loads of data another CU just wrote), so many short levels
// beat few long ones: the optimum at N=5001 is 5 levels of 4-5 nodes.  A level that fits TOPW segments could run with
// the rest of the tree inside ONE workgroup (bt_top_kernel); measured on MI355X this only pays for the root level
// (the inter-level latency is memory round trips, not launch overhead), hence TOPW = 1.
// seg_len[0..1] > 0 pin the segment length of levels 0 / 1 (tests, tuning).
// twisted: plan for the two-sided elimination (a segment of m nodes costs m/2+1 steps; odd lengths, at most BS_PAR_MAX,
// waste nothing).  The plan is marked twisted only if every level below the root qualifies.
int plan_levels(int N, const int seg_len[2], SolvePlan& best, bool twisted = false) {
    const double t_node = 2.3, t_launch = 4.0;
    double best_cost = 1e300;
    best.nl = 0;
    best.twisted = 0;
    // (cand 1: graphs too long for MAXL twisted levels of equal length -- beyond ~130 000 nodes -- with the twisted maximum on every
    //  level and whatever is left, a dozen nodes, as a one-sided root: at N = 300 007 segments of 8 on the one-sided kernels cost 8 node
    //  steps each, segments of 7 on the twisted ones 4 -- 1.40 -> 1.1x ms per LM iteration)
    for (int depth = 1; depth <= MAXL; ++depth)
    for (int cand = 0; cand < 2; ++cand) {
        int m_auto = std::max(4, (int)std::ceil(std::pow((double)N, 1.0 / depth)) - 1);
        if (cand == 1) {
            if (!(twisted && depth == MAXL && m_auto > BS_PAR_MAX)) continue;
            m_auto = BS_PAR_MAX;
        } else if (twisted) {
            if (m_auto > BS_PAR_MAX && depth < MAXL) continue;          // a deeper tree reaches a length the twisted path handles
            if (m_auto % 2 == 0 && m_auto + 1 <= BS_PAR_MAX) ++m_auto;
        }
        SolvePlan c;
        c.nl = 0;
        int n = N;
        bool tw = twisted;
        for (int l = 0; l < MAXL; ++l) {
            LevelPlan L;
            L.n = n;
            int m = m_auto;
            if (seg_len && l < 2 && seg_len[l] > 0) m = std::max(seg_len[l], 4);
            if (l == MAXL - 1 || l >= depth - 1 || m + 1 >= n || n <= (twisted ? BS_PAR_MAX : 12)) { L.m = n; L.P = 1; L.nsep = 0; c.lv[c.nl++] = L; break; }
            if (m > BS_PAR_MAX) tw = false;
            L.m = m; L.P = (n + m) / (m + 1); L.nsep = n / (m + 1);
            c.lv[c.nl++] = L;
            n = L.nsep;
        }
        c.top = c.nl - 1;
        while (c.top > 0 && c.lv[c.top - 1].P <= TOPW && (c.nl - (c.top - 1)) <= MAXTOP) --c.top;
        if (c.nl < 2 || c.top != c.nl - 1) tw = false;                  // twisted levels exist only on the down-sweep path
        c.twisted = tw ? 1 : 0;
        double cost = t_launch;
        for (int l = 0; l < c.nl; ++l) {
            const bool root = l == c.nl - 1;
            const bool ltw = tw && (!root || c.lv[l].n <= BS_PAR_MAX);
            // (a level of more segments than the chip holds at once runs in rounds: 256 CUs x 3 workgroups of the level kernels)
            const double rounds = c.lv[l].P > 3072 ? c.lv[l].P / 768.0 : 1.0;
            cost += rounds * segment_steps(c.lv[l].m, ltw) * t_node + (l < c.top ? 2 * t_launch : 0.0);
        }
        if (cost < best_cost - 1e-9) { best_cost = cost; best = c; }
    }
    return best.nl;
}

struct LevelBufs { double *fac, *inv, *Dsep, *rsep, *cL, *cR, *cgL, *cgR, *fill, *x, *gx; size_t prod_bytes; };     // gx: influence matrices handed down (levels >= 1); prod_bytes: Dsep .. cgR, contiguous

struct Workspace {
    double *lin, *loss_part, *part, *Hd, *Ho, *rhs, *dx, *nodes_t, *vels_t, *state;
    double *lin2, *Hd2, *Ho2, *rhs2;          // second linearisation buffer (the trial point = the next step, if accepted)
    double *red, *red2;                       // reprojection factor: per-link reductions (same double buffering)
    int* ready;                               // down-sweep: one word per segment of every level (+ the root)
    size_t ready_bytes;
    int* flags;
    LevelBufs lv[MAXL];
    size_t bytes;
};

// carve the workspace; sizes use worst-case level shapes (segment length >= 4: level l has at most N / 5^l + 2 nodes)
Workspace carve(void* base, int N) {
    Workspace w;
    char* p = (char*)base;
    auto take = [&](size_t nd) { double* r = (double*)p; p += align_up(nd * sizeof(double)); return r; };
    const int M = std::max(N - 1, 1);
    const int nblk = (M + 63) / 64;
    w.lin = take((size_t)LIN_C * M);
    const int nlb = (N + LB_NODES - 1) / LB_NODES;     // workgroups of linbuild / trial_lin (>= nblk)
    w.loss_part = take(std::max(nblk, nlb) + 2);
    w.part = take(2 * (size_t)std::max(std::max(nblk, nlb), 1024) + 2);           // (trial_elim_kernel: one pair per workgroup, at most one workgroup per CU)
    w.Hd = take((size_t)N * 81);
    w.Ho = take((size_t)N * 81);
    w.rhs = take((size_t)N * 9);
    w.lin2 = take((size_t)LIN_C * M);
    w.Hd2 = take((size_t)N * 81);
    w.Ho2 = take((size_t)N * 81);
    w.rhs2 = take((size_t)N * 9);
    w.red = take((size_t)M * RP_REC);
    w.red2 = take((size_t)M * RP_REC);
    w.dx = take((size_t)N * 9);
    w.nodes_t = take((size_t)N * 7);
    w.vels_t = take((size_t)N * 3);
    w.state = take(STATE_DOUBLES);
    w.flags = (int*)take(4);         // [0] solver error of the launched levels, [2] ticket, [4], [5] solver error of the fused level-0 elimination (by parity)
    w.ready_bytes = align_up(((size_t)N / 3 + 64 * MAXL) * READY_STRIDE * sizeof(int));   // segments of all levels < N/4 + ...
    w.ready = (int*)take(w.ready_bytes / sizeof(double));
    int n = N;
    for (int l = 0; l < MAXL; ++l) {
        LevelBufs& b = w.lv[l];
        int segs = n / 5 + 2;     // m >= 4 -> stride >= 5
        b.fac = take((size_t)n * FAC);
        b.inv = take((size_t)n * 9);
        b.x = take((size_t)n * 9);
        b.Dsep = take((size_t)segs * 81);
        b.rsep = take((size_t)segs * 9);
        b.cL = take((size_t)segs * 81);
        b.cR = take((size_t)segs * 81);
        b.fill = take((size_t)segs * 81);
        b.cgL = take((size_t)segs * 9);
        b.cgR = take((size_t)segs * 9);
        b.prod_bytes = (size_t)(p - (char*)b.Dsep);
        b.gx = l >= 1 ? take((size_t)n * 171) : nullptr;
        n = segs;
    }
    w.bytes = (size_t)(p - (char*)base);
    return w;
}

static LevelDst level_dst(const LevelBufs& b, double* x) {
    LevelDst d{};
    d.fac = b.fac; d.inv = b.inv; d.Dsep = b.Dsep; d.rsep = b.rsep; d.cL = b.cL; d.cR = b.cR; d.cgL = b.cgL; d.cgR = b.cgR;
    d.fill = b.fill; d.x = x;
    return d;
}
static LevelSrc level_src_from(const LevelBufs& pb, int Pprev) {
    LevelSrc s{};
    s.level0 = 0; s.Dsep = pb.Dsep; s.rsep = pb.rsep; s.cL = pb.cL; s.cR = pb.cR; s.cgL = pb.cgL; s.cgR = pb.cgR;
    s.fill = pb.fill; s.Pprev = Pprev;
    return s;
}

// Pairs of down-sweep levels share one hand-off (SweepLevel::merge) unless ISLAM_PVGO_NO_MERGE=1 (A/B runs): 70.4 vs 71.5-71.8 us
// per LM iteration at N = 5001.  History: composing the two levels' influence matrices into one 9 x 28 map per lane (342 LDS
// reads + FMAs) was slower (73.0 us); fetching rows of the producer's G lane by lane (19 stores / 19 loads, 152 bytes apart
// between lanes) made the exchange end 2-3 us after the upstream words were published (no gain); with the matrices copied through
// LDS in contiguous 512-byte stores / loads the exchange is done ~1 us after the influence matrices are.
static bool merge_levels() {
    static const bool v = [] { const char* e = std::getenv("ISLAM_PVGO_NO_MERGE"); return !(e && e[0] == '1'); }();
    return v;
}

// every down-sweep launch of the process gets its own serial number: what its ready words must hold to count as published
static int next_serial() {
    static std::atomic<int> g_serial{0};
    int serial = ++g_serial;
    if (serial == 0) serial = ++g_serial;                     // 0 is the reset value of the ready words
    return serial;
}

static void launch_tw(const LevelSrc& src, const LevelDst& dst, int n, int m, int* flags, int seg0, int nseg, Gate gate, hipStream_t s) {
    if (src.level0) hipLaunchKernelGGL(bt_eliminate_tw_kernel<1>, dim3(xcd_grid(nseg)), dim3(192), 0, s, src, dst, n, m, flags, seg0, nseg, gate);
    else hipLaunchKernelGGL(bt_eliminate_tw_kernel<0>, dim3(xcd_grid(nseg)), dim3(192), 0, s, src, dst, n, m, flags, seg0, nseg, gate);
}

// the up-sweep launch of one level below the root: one workgroup per segment (three wavefronts when twisted)
static void launch_eliminate(const LevelPlan& L, bool tw, const LevelSrc& src, const LevelDst& dst, int* flags, hipStream_t s,
                             Gate gate) {
    if (tw)
        launch_tw(src, dst, L.n, L.m, flags, 0, L.P, gate, s);
    else
        hipLaunchKernelGGL(bt_eliminate_kernel, dim3(xcd_grid(L.P)), dim3(64), 0, s, src, dst, L.n, L.m, flags, 0, L.P, gate);
}

// Enqueue levels [lbegin, nl): `first` describes the source of level lbegin (level-0 arrays, or the level-0 products when
// lbegin == 1), xout receives the solution of level lbegin.  Big levels: one launch each way; levels >= sp.top: one launch.
// skip_first: the elimination of level lbegin has been enqueued by the caller (trial_elim_kernel eliminates level 0 itself).
int enqueue_levels(const Workspace& w, const SolvePlan& sp, int lbegin, const LevelSrc& first, const LevelBufs* first_prev,
                   double* xout, int* flags, hipStream_t s, hipEvent_t* evs, int* nev, Gate gate = Gate{nullptr, 0.0},
                   bool skip_first = false) {
    static bool lds_attr_set[64] = {};                       // per device: the attribute lives in the device's code object
    int dev_i = 0;
    ISLAM_HIP_CHECK(hipGetDevice(&dev_i));
    if (dev_i >= 0 && dev_i < 64 && !lds_attr_set[dev_i]) {
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)bt_top_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            TOPW * LDS_PER_WAVE * (int)sizeof(double)));
        lds_attr_set[dev_i] = true;
    }
    const int nl = sp.nl, top = std::max(sp.top, lbegin);
    int ne = 0;
    if (evs) (void)hipEventRecord(evs[ne++], s);
    auto src_of = [&](int l) {
        if (l == lbegin) return first;
        const LevelBufs& pb = (l - 1 == lbegin - 1 && first_prev) ? *first_prev : w.lv[l - 1];
        return level_src_from(pb, sp.lv[l - 1].P);
    };
    auto x_of = [&](int l) { return l == lbegin ? xout : w.lv[l].x; };
    const bool sweep = (top == nl - 1) && (top > lbegin);      // root alone in the top kernel, at least one level below
    const bool tw = sp.twisted && sweep && lbegin == 0;
    for (int l = lbegin; l < top; ++l) {
        if (l == lbegin && skip_first) continue;
        launch_eliminate(sp.lv[l], tw, src_of(l), level_dst(w.lv[l], x_of(l)), flags, s, gate);
        if (evs) (void)hipEventRecord(evs[ne++], s);
    }
    if (sweep) {
        const int serial = next_serial();
        SweepArgs a{};
        a.root_src = src_of(top);
        a.root_dst = level_dst(w.lv[top], x_of(top));
        a.root_n = sp.lv[top].n;
        a.ready = w.ready;
        a.serial = serial;
        a.root_twisted = (tw && sp.lv[top].n <= BS_PAR_MAX) ? 1 : 0;
        a.outer_x = nullptr;
        a.outer_flag = 0;
        a.nl = top - lbegin;
        int flag = 1, blk = 8;
        for (int i = 0; i < a.nl; ++i) {
            const int l = top - 1 - i;
            SweepLevel& L = a.lv[i];
            L.fac = w.lv[l].fac; L.inv = w.lv[l].inv; L.xsep = x_of(l + 1); L.x = x_of(l);
            L.n = sp.lv[l].n; L.m = sp.lv[l].m; L.P = sp.lv[l].P;
            L.seg0 = 0; L.nseg = L.P; L.twisted = tw ? 1 : 0; L.outer = 0; L.store_left = 0; L.x_last = L.n * 9 - 1;
            L.merge = 0; L.publish_g = 0; L.skip_x = 0; L.gflag0 = 0; L.gx = nullptr;
            L.flag0 = flag;
            L.up_flag0 = i == 0 ? 0 : a.lv[i - 1].flag0;
            L.up_stride = i == 0 ? (1 << 30) : sp.lv[l + 1].m + 1;
            a.first_block[i] = blk;
            flag += L.P;
            blk += xcd_grid(L.P);
        }
        a.first_block[a.nl] = blk;
        // pairs of levels share ONE hand-off, from the bottom of the tree up: (L0, L1), (L2, L3), ... (twisted levels with
        // influence matrices only, i.e. a fully resident grid)
        if (tw && blk <= 2048 && merge_levels()) {
            for (int i = a.nl - 1; i >= 1; i -= 2) {
                SweepLevel &C = a.lv[i], &Pp = a.lv[i - 1];
                const int lp = top - 1 - (i - 1);            // tree level of the producer
                if (C.m > BS_PAR_MAX || Pp.m > BS_PAR_MAX || w.lv[lp].gx == nullptr) continue;
                C.merge = 1;
                Pp.publish_g = 1;
                Pp.skip_x = 1;                               // (its x array is only ever read as the consumer's separators)
                Pp.gx = w.lv[lp].gx;
                Pp.gflag0 = flag;
                flag += Pp.P;
            }
        }
        if ((size_t)flag * READY_STRIDE * sizeof(int) > w.ready_bytes) return fail(ISLAM_EARG, "pvgo: ready-flag buffer too small (%d words)", flag);
        hipLaunchKernelGGL(bt_downsweep_kernel, dim3(blk), dim3(128), 0, s, a, flags, gate);
        if (evs) (void)hipEventRecord(evs[ne++], s);
    } else {
        TopArgs a{};
        a.nl = nl - top;
        int maxP = 1;
        for (int i = 0; i < a.nl; ++i) {
            const int l = top + i;
            a.src[i] = src_of(l);
            a.dst[i] = level_dst(w.lv[l], x_of(l));
            a.n[i] = sp.lv[l].n; a.m[i] = sp.lv[l].m; a.P[i] = sp.lv[l].P;
            maxP = std::max(maxP, sp.lv[l].P);
        }
        hipLaunchKernelGGL(bt_top_kernel, dim3(1), dim3(64 * maxP), maxP * LDS_PER_WAVE * sizeof(double), s, a, flags, gate);
        if (evs) (void)hipEventRecord(evs[ne++], s);
        for (int l = top - 1; l >= lbegin; --l) {
            hipLaunchKernelGGL(bt_backsub_kernel, dim3(xcd_grid(sp.lv[l].P)), dim3(64), 0, s, w.lv[l].fac, w.lv[l].inv, x_of(l + 1),
                               x_of(l), sp.lv[l].n, sp.lv[l].m, 0, sp.lv[l].P, gate);
            if (evs) (void)hipEventRecord(evs[ne++], s);
        }
    }
    if (nev) *nev = ne;
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

// state + flags + ready words of a run: one launch while the ready words fit a single pass of 64 workgroups (graphs up to ~20 000
// nodes), the copy engine's fill + the one-wave kernel beyond
static int enqueue_control_init(const Workspace& w, const islam_pvgo_params* prm, hipStream_t s) {
    if (w.ready_bytes <= (1u << 20)) {
        const unsigned n16 = (unsigned)(w.ready_bytes / 16);
        hipLaunchKernelGGL(control_init_kernel, dim3(std::max(1u, std::min(64u, (n16 + 255u) / 256u))), dim3(256), 0, s, w.state, w.flags,
                           prm->radius, prm->down, reinterpret_cast<uint4*>(w.ready), n16);
    } else {
        ISLAM_HIP_CHECK(hipMemsetAsync(w.ready, 0, w.ready_bytes, s));
        hipLaunchKernelGGL(control_init_kernel, dim3(1), dim3(64), 0, s, w.state, w.flags, prm->radius, prm->down, (uint4*)nullptr, 0u);
    }
    return ISLAM_OK;
}

// single-GPU solves use the twisted elimination; ISLAM_PVGO_ONESIDED=1 keeps the one-sided path (A/B measurements)
static bool solve_twisted() {
    static const bool tw = [] { const char* e = std::getenv("ISLAM_PVGO_ONESIDED"); return !(e && e[0] == '1'); }();
    return tw;
}

// enqueue one damped solve: Hd.diag += Hd.diag*damping; dx = A^-1 rhs
int enqueue_solve(const Workspace& w, double* Hd, const double* Ho, const double* rhs, const double* state,
                  double damping, int N, const int seg_len[2], double* dx, hipStream_t s, hipEvent_t* evs = nullptr,
                  int* nev = nullptr, Gate gate = Gate{nullptr, 0.0}) {
    SolvePlan sp;
    plan_levels(N, seg_len, sp, solve_twisted());
    LevelSrc src{};
    src.level0 = 1; src.Hd = Hd; src.Ho = Ho; src.rhs0 = rhs; src.state = state; src.damping_override = damping;
    return enqueue_levels(w, sp, 0, src, nullptr, dx, w.flags, s, evs, nev, gate);
}

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

namespace islam {

// Up-sweep of levels 0 .. xl over the rank's own segments.  Hd/Ho/rhs: LOCAL level-0 arrays whose row 0 is global node
// `node0`; exchange: 351*P_xl doubles (array-major like the level-0 products), own rows written -- and the rank's share of its
// left outer separator's block, a row of the previous rank's segment; zero_exchange: every other row is zeroed first (a caller
// that never lets anything else touch the buffer zeroes it once and passes false).
int shard_upsweep_gated(double* Hd, const double* Ho, const double* rhs, double damping, const double* state, int N,
                        const int seg_len[2], int world, int rank, int node0, void* workspace, size_t workspace_bytes,
                        double* exchange, bool zero_exchange, int* flags, Gate gate, hipStream_t s) {
    SolvePlan sp;
    const int nl = shard_plan(N, seg_len, sp);
    ShardRanges R;
    if (nl < 2 || shard_ranges(sp, world, rank, R) != 0) return fail(ISLAM_EARG, "islam_pvgo_shard_upsweep: N=%d, world=%d, rank=%d", N, world, rank);
    if (workspace_bytes < islam_pvgo_workspace_bytes(N)) return fail(ISLAM_EARG, "islam_pvgo_shard_upsweep: workspace too small");
    Workspace w = carve((void*)align_up((size_t)workspace), N);
    if (zero_exchange) ISLAM_HIP_CHECK(hipMemsetAsync(exchange, 0, sizeof(double) * 351 * (size_t)sp.lv[R.xl].P, s));
    LevelBufs xb{};
    products_view(exchange, sp.lv[R.xl].P, xb);
    for (int l = 0; l <= R.xl; ++l) {
        LevelSrc src{};
        if (l == 0) {
            src.level0 = 1;
            src.Hd = Hd - (ptrdiff_t)node0 * 81; src.Ho = Ho - (ptrdiff_t)node0 * 81; src.rhs0 = rhs - (ptrdiff_t)node0 * 9;
            src.state = state; src.damping_override = damping;
        } else {
            src = level_src_from(w.lv[l - 1], sp.lv[l - 1].P);
        }
        LevelBufs ob = w.lv[l];
        if (l == R.xl) { ob.Dsep = xb.Dsep; ob.rsep = xb.rsep; ob.cL = xb.cL; ob.cR = xb.cR; ob.fill = xb.fill; ob.cgL = xb.cgL; ob.cgR = xb.cgR; }
        if (sp.twisted)
            launch_tw(src, level_dst(ob, nullptr), sp.lv[l].n, sp.lv[l].m, flags, R.seg0[l], R.nseg[l], gate, s);
        else
            hipLaunchKernelGGL(bt_eliminate_kernel, dim3(xcd_grid(R.nseg[l])), dim3(64), 0, s, src, level_dst(ob, nullptr), sp.lv[l].n,
                               sp.lv[l].m, flags, R.seg0[l], R.nseg[l], gate);
    }
    if (R.seg0[R.xl] > 0 && R.xl > 0) {                      // this rank's share of its LEFT outer separator's block (see OuterFix)
        OuterFix f{};
        f.n = R.xl;
        for (int l = 0; l < R.xl; ++l) { f.cL[l] = w.lv[l].cL + (size_t)R.seg0[l] * 81; f.cgL[l] = w.lv[l].cgL + (size_t)R.seg0[l] * 9; }
        f.Dsep = xb.Dsep + (size_t)(R.seg0[R.xl] - 1) * 81;
        f.rsep = xb.rsep + (size_t)(R.seg0[R.xl] - 1) * 9;
        hipLaunchKernelGGL(outer_block_kernel, dim3(1), dim3(128), 0, s, f, gate);
    }
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

// Levels above xl from the SUMMED exchange buffer (redundantly on every rank: one elimination launch each), then ONE launch
// (bt_downsweep_kernel) for the root and the whole back-substitution: the replicated levels in full, levels xl .. 0 over the
// rank's own segments.  dx: LOCAL array, row 0 = global node `node0`; rows node0 .. the rank's right outer separator are
// written (the left outer separator's row too when the rank has one).
static int shard_downsweep_planned(const SolvePlan& sp, const ShardRanges& R, const Workspace& w, const double* exchange, int world, int node0,
                                   double* dx, int* flags, Gate gate, hipStream_t s, const double* fwd_src = nullptr, double* fwd_dst = nullptr);
int shard_downsweep_gated(const double* exchange, int N, const int seg_len[2], int world, int rank, int node0, void* workspace,
                          size_t workspace_bytes, double* dx, int* flags, Gate gate, hipStream_t s) {
    SolvePlan sp;
    const int nl = shard_plan(N, seg_len, sp);
    ShardRanges R;
    if (nl < 2 || shard_ranges(sp, world, rank, R) != 0) return fail(ISLAM_EARG, "islam_pvgo_shard_downsweep: N=%d, world=%d, rank=%d", N, world, rank);
    if (workspace_bytes < islam_pvgo_workspace_bytes(N)) return fail(ISLAM_EARG, "islam_pvgo_shard_downsweep: workspace too small");
    Workspace w = carve((void*)align_up((size_t)workspace), N);
    return shard_downsweep_planned(sp, R, w, exchange, world, node0, dx, flags, gate, s);
}
// (the plan, the ranges and the carved workspace do not change within a run: the loop in the library computes them once)
static int shard_downsweep_planned(const SolvePlan& sp, const ShardRanges& R, const Workspace& w, const double* exchange, int world, int node0,
                                   double* dx, int* flags, Gate gate, hipStream_t s, const double* fwd_src, double* fwd_dst) {
    const int nl = sp.nl;
    const int top = nl - 1, xl = R.xl;
    const bool tw = sp.twisted != 0;
    LevelBufs pb{};
    products_view(const_cast<double*>(exchange), sp.lv[xl].P, pb);
    auto src_of = [&](int l) { return level_src_from(l == xl + 1 ? pb : w.lv[l - 1], sp.lv[l - 1].P); };
    double* x0 = dx - (ptrdiff_t)node0 * 9;                                   // level-0 solution, global indexing
    auto x_of = [&](int l) { return l == 0 ? x0 : w.lv[l].x; };
    for (int l = xl + 1; l < top; ++l)
        launch_eliminate(sp.lv[l], tw, src_of(l), level_dst(w.lv[l], x_of(l)), flags, s, gate);
    const int serial = next_serial();
    SweepArgs a{};
    a.root_src = src_of(top);
    a.root_dst = level_dst(w.lv[top], x_of(top));
    a.root_n = sp.lv[top].n;
    a.ready = w.ready;
    a.serial = serial;
    a.root_twisted = (tw && sp.lv[top].n <= BS_PAR_MAX) ? 1 : 0;
    a.nl = top;
    a.outer_x = nullptr;
    a.outer_flag = 0;
    a.fwd_src = fwd_src; a.fwd_dst = fwd_dst;
    int flag = 1, blk = 8;
    for (int i = 0; i < a.nl; ++i) {
        const int l = top - 1 - i;
        SweepLevel& L = a.lv[i];
        L.fac = w.lv[l].fac; L.inv = w.lv[l].inv; L.xsep = x_of(l + 1); L.x = x_of(l);
        L.n = sp.lv[l].n; L.m = sp.lv[l].m; L.P = sp.lv[l].P;
        const bool local = l <= xl;
        L.seg0 = local ? R.seg0[l] : 0;
        L.nseg = local ? R.nseg[l] : L.P;
        L.twisted = tw ? 1 : 0;
        L.outer = (l < xl && R.seg0[l] > 0) ? 1 : 0;
        L.store_left = (local && R.seg0[l] > 0) ? 1 : 0;
        L.x_last = std::min(L.n, (L.seg0 + L.nseg) * (L.m + 1)) * 9 - 1;
        L.flag0 = flag;
        L.up_flag0 = i == 0 ? 0 : a.lv[i - 1].flag0;
        L.up_stride = i == 0 ? (1 << 30) : sp.lv[l + 1].m + 1;
        a.first_block[i] = blk;
        flag += L.P;
        blk += xcd_grid(L.nseg);
        if (l == xl + 1 && R.seg0[xl] > 0) {                // the level that solves the rank's left cut node (node seg0[xl]-1 there)
            a.outer_x = x_of(l) + (size_t)(R.seg0[xl] - 1) * 9;
            a.outer_flag = L.flag0 + (R.seg0[xl] - 1) / (L.m + 1);
        }
    }
    if (xl + 1 == top && R.seg0[xl] > 0) { a.outer_x = x_of(top) + (size_t)(R.seg0[xl] - 1) * 9; a.outer_flag = 0; }
    a.first_block[a.nl] = blk;
    if (world == 1 && tw && blk <= 2048 && merge_levels()) {        // (one rank: every level in full -- the pairing of enqueue_levels)
        for (int i = a.nl - 1; i >= 1; i -= 2) {
            SweepLevel &C = a.lv[i], &Pp = a.lv[i - 1];
            const int lp = top - 1 - (i - 1);
            if (C.m > BS_PAR_MAX || Pp.m > BS_PAR_MAX || w.lv[lp].gx == nullptr) continue;
            C.merge = 1;
            Pp.publish_g = 1;
            Pp.skip_x = 1;
            Pp.gx = w.lv[lp].gx;
            Pp.gflag0 = flag;
            flag += Pp.P;
        }
    }
    if ((size_t)flag * READY_STRIDE * sizeof(int) > w.ready_bytes) return fail(ISLAM_EARG, "pvgo: ready-flag buffer too small (%d words)", flag);
    hipLaunchKernelGGL(bt_downsweep_kernel, dim3(blk), dim3(128), 0, s, a, flags, gate);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

// ---- the sharded loop on the fused kernel -----------------------------------------------------------
// One trial of a rank = trial_elim_kernel over the rank's own level-0 segments (trial step, linearisation at the trial point,
// level-0 elimination under the speculated damping) -> levels 1 .. xl -> shard_pack_kernel -> ONE all-reduce of
//   [ interface blocks of the NEXT solve (351 per segment of the exchange level) | sum r^2 | sum JD.(2R+JD) | failed pivots |
//     per cut: the two ranks' parts of the cut node's raw diagonal (9 + 9) ]
// -> shard_decide_kernel (LM decision of THIS trial, replicated on every rank; validates the speculation; clamps the cut nodes'
// diagonals) -> levels above xl + root + local back-substitution (bt_downsweep_kernel) -> dx of the next trial.
//
// What makes one collective enough: a rank never needs anything of a node it does not hold.  Its stretch ends AT its right outer
// separator (node sR): the link sR -> sR+1 belongs to the next rank, which holds both of its nodes (it back-substitutes its left
// outer separator itself, from the replicated top of the tree) and hands the link's part of node sR's block and right-hand side
// up in the exchange buffer, next to the Schur parts it has always handed up (OuterFix).  No halo rows, no second all-reduce.
// PyPose clamps the diagonal of A before damping it (A.diagonal().clamp_): the cut nodes' diagonals are sums over two ranks, so
// both raw parts travel in the message and every rank applies the clamp to the sum (a no-op unless an entry leaves
// [vmin, vmax] -- then the damped difference is added to the row of the exchange buffer).
struct PackArgs {
    OuterFix f;                   // Schur parts of the left outer separator (n may be 0)
    int has_left, has_right;
    const double* share;          // trial_elim_kernel's FusedArgs::share of the linearisation being solved
    const double* Hd_right;       // block of the rank's right outer separator in that linearisation (undamped)
    const double* st; TRParams tr;
    int damp_mode;                // 0: speculated_damping(st) once; 1: the list of the current linearisation; 2: st[2] once (first solve)
    const double* part; int nwg;  // trial sums (nullptr: a solve without a trial -- the scalars stay zero)
    double* msg; int nmsg, rank;
};
__device__ __forceinline__ double shard_damp(const double* st, const TRParams& tr, int mode, double v) {
    if (mode == 1) {
        const int n = (int)st[8] + 1;
        for (int i = 0; i < n; ++i) v = v + v * st[STATE_HIST + i];
        return v;
    }
    const double d = mode == 0 ? speculated_damping(st, tr) : st[2];
    return v + v * d;
}
__device__ __forceinline__ void shard_pack(const PackArgs& a) {
    const int t = threadIdx.x;
    for (int i = t; i < a.nmsg; i += 128) a.msg[i] = 0.0;
    __syncthreads();
    if (t < 90 && a.has_left) {
        double v = 0.0;
        for (int i = 0; i < a.f.n; ++i) v += t < 81 ? a.f.cL[i][t] : a.f.cgL[i][t - 81];
        double sh = a.share[t];
        if (t < 81 && t % 10 == 0) {
            a.msg[3 + 18 * (a.rank - 1) + 9 + t / 10] = sh;
            sh = shard_damp(a.st, a.tr, a.damp_mode, sh);
        }
        if (t < 81) a.f.Dsep[t] = sh - v; else a.f.rsep[t - 81] = sh - v;
    }
    if (t >= 96 && t < 105 && a.has_right) a.msg[3 + 18 * a.rank + (t - 96)] = a.Hd_right[(t - 96) * 10];
    if (t >= 64 && a.part) {                                      // wave 1: the rank's sums, in index order
        const int lane = t - 64;
        double ssum = 0.0, qsum = 0.0;
        for (int i = lane; i < a.nwg; i += 64) { ssum += a.part[2 * i]; qsum += a.part[2 * i + 1]; }
        ssum = wave_sum(ssum);
        qsum = wave_sum(qsum);
        if (lane == 0) { a.msg[0] = ssum; a.msg[1] = qsum; a.msg[2] = a.part[2 * a.nwg]; }
    }
}
__global__ __launch_bounds__(128) void shard_pack_kernel(PackArgs a, Gate gate) {
    if (gate_closed(gate)) return;
    shard_pack(a);
}

// mode 0: a trial (LM decision; d_spec = the damping the solve that is already eliminated used); 1: a solve without a trial
// (only the clamp); 2: the first linearisation (its loss opens the run)
struct DecideArgs {
    const double* msg; double* ex_Dsep; double* st; TRParams tr; double* report; double seq; int mode, damp_mode, world, Pxl;
    int plain_report;             // report is device memory the down-sweep forwards (lm_control)
    double vmin, vmax;
};
// (the state block is staged through LDS: lm_control is a chain of ~40 dependent reads and writes of it -- 3 us of global-memory round
// trips on one lane when it works on the device copy, and this kernel sits on the critical path of every trial)
__device__ __forceinline__ void shard_decide(const DecideArgs& a, double* st_l) {
    const int t = threadIdx.x;
    if (t < STATE_DOUBLES) st_l[t] = a.st[t];
    __syncthreads();
    const double d_spec = a.mode == 0 ? speculated_damping(st_l, a.tr) : -1.0;
    for (int i = t; i < 9 * (a.world - 1); i += (int)blockDim.x) {
        const int b = i / 9, j = i - 9 * b;
        const double da = a.msg[3 + 18 * b + j], db = a.msg[3 + 18 * b + 9 + j], tot = da + db;
        const double cl = fmin(fmax(tot, a.vmin), a.vmax);
        if (cl != tot) {
            const int slot = (int)((long long)(b + 1) * a.Pxl / a.world) - 1;          // last exchange-level segment of rank b
            a.ex_Dsep[(size_t)slot * 81 + j * 10] += shard_damp(st_l, a.tr, a.damp_mode, cl) -
                                                     (shard_damp(st_l, a.tr, a.damp_mode, da) + shard_damp(st_l, a.tr, a.damp_mode, db));
        }
    }
    __syncthreads();
    if (t == 0) {
        if (a.mode == 2) { st_l[0] = a.msg[0]; st_l[1] = a.msg[0]; st_l[8] = 0.0; st_l[11] = 1.0; st_l[12] = 0.0; st_l[13] = 0.0; }
        else if (a.mode == 0) lm_control(a.msg[0], a.msg[1], st_l, a.msg[2] > 0.0, a.tr, a.report, a.seq, d_spec, a.plain_report != 0);
    }
    __syncthreads();
    if (t < STATE_DOUBLES && a.mode != 1) a.st[t] = st_l[t];
}
__global__ __launch_bounds__(64) void shard_decide_kernel(DecideArgs a, Gate gate) {
    __shared__ double st_l[STATE_DOUBLES];
    if (gate_closed(gate)) return;
    shard_decide(a, st_l);
}
// one rank: nothing to sum between the two, no cut -- the rank's sums go from the partials to the decision through LDS (the round trip
// of the three scalars through the message in global memory and two of the barriers cost ~1.5 us of a launch that every trial waits for)
__global__ __launch_bounds__(128) void shard_pack_decide_kernel(PackArgs p, DecideArgs d, Gate gate) {
    __shared__ double st_l[STATE_DOUBLES];
    __shared__ double m3[3];
    if (gate_closed(gate)) return;
    const int t = threadIdx.x;
    if (t < STATE_DOUBLES) st_l[t] = d.st[t];
    if (t >= 64) {
        const int lane = t - 64;
        double ssum = 0.0, qsum = 0.0;
        for (int i = lane; i < p.nwg; i += 64) { ssum += p.part[2 * i]; qsum += p.part[2 * i + 1]; }
        ssum = wave_sum(ssum);
        qsum = wave_sum(qsum);
        if (lane == 0) { m3[0] = ssum; m3[1] = qsum; m3[2] = p.part[2 * p.nwg]; }
    }
    __syncthreads();
    if (t == 0) {
        if (d.mode == 2) { st_l[0] = m3[0]; st_l[1] = m3[0]; st_l[8] = 0.0; st_l[11] = 1.0; st_l[12] = 0.0; st_l[13] = 0.0; }
        else lm_control(m3[0], m3[1], st_l, m3[2] > 0.0, d.tr, d.report, d.seq, speculated_damping(st_l, d.tr), d.plain_report != 0);
    }
    __syncthreads();
    if (t < STATE_DOUBLES) d.st[t] = st_l[t];
}

__global__ void shard_close_gate_kernel(double* __restrict__ st) {
    if (threadIdx.x == 0) st[14] = -1.0;
}

size_t shard_fused_scratch_doubles(int N, int world) {
    const size_t n = (size_t)N + 2, ex = 351 * (n / 5 + 2) + 3 + 18 * (size_t)world;
    auto a256 = [](size_t k) { return align_up(k * sizeof(double)) / sizeof(double); };
    return 2 * a256(ex) + 2 * a256(96) + a256(32) + 64;
}

int run_chain_sharded_fused(const ShardSum& red, int world, int rank, double* nodes, double* vels, const double* poses, const double* drots,
                            const double* dtrans, const double* dvels, const double* dts, int N, const islam_pvgo_params* prm,
                            void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes, islam_pvgo_result* res,
                            long long* exchanged_bytes, hipStream_t s, int* taken, const double** out_nodes, const double** out_vels,
                            int* own0, int* own1) {
    *taken = 0;
    static const bool off = [] { const char* e = std::getenv("ISLAM_SHARD_FUSED"); return e && e[0] == '0'; }();
    SolvePlan sp;
    const int nl = shard_plan(N, prm->seg_len, sp);
    ShardRanges R;
    if (off || nl < 2 || shard_ranges(sp, world, rank, R) != 0) return ISLAM_OK;
    static const int fz_spare = [] { const char* e = std::getenv("ISLAM_FZ_SPARE"); return e ? std::atoi(e) : 16; }();
    // (the same plans the single-GPU loop fuses, decided on numbers every rank shares: all ranks take the same path)
    int max_nseg = 0;
    for (int r = 0; r < world; ++r) {
        ShardRanges Rr;
        if (shard_ranges(sp, world, r, Rr) != 0) return ISLAM_OK;
        max_nseg = std::max(max_nseg, Rr.nseg[0]);
        if (Rr.nseg[0] < 1) return ISLAM_OK;
    }
    const int cus = std::max(device_cus() - fz_spare, 1);
    if (!(N > 96 && sp.twisted && sp.top == sp.nl - 1 && sp.lv[0].m <= FZ_MAXM && prm->reject < STATE_DOUBLES - STATE_HIST - 1 &&
          (max_nseg + std::min(max_nseg, cus) - 1) / std::min(max_nseg, cus) <= FZ_S))
        return ISLAM_OK;
    if (workspace_bytes < islam_pvgo_workspace_bytes(N)) return fail(ISLAM_EARG, "islam_pvgo_run_chain_sharded: workspace too small");
    if (scratch_bytes < shard_fused_scratch_doubles(N, world) * sizeof(double)) return fail(ISLAM_EARG, "islam_pvgo_run_chain_sharded: scratch too small");
    *taken = 1;
    const int M = N - 1, xl = R.xl, Pxl = sp.lv[xl].P, m = sp.lv[0].m, stride = m + 1;
    const int seg_lo = R.seg0[0], nseg = R.nseg[0], first_node = seg_lo * stride, sR = (seg_lo + nseg - 1) * stride + m;
    const bool has_left = seg_lo > 0, has_right = sR < N - 1;
    const int N_eff = has_right ? sR + 1 : N;
    const int nwg = std::min(nseg, cus);
    const int nmsg = 3 + 18 * (world - 1);
    const size_t nex = 351 * (size_t)Pxl + nmsg;
    Workspace w = carve((void*)align_up((size_t)workspace), N);
    double* p = (double*)align_up((size_t)scratch);
    auto take = [&](size_t k) { double* r = p; p += align_up(k * sizeof(double)) / sizeof(double); return r; };
    double* ex_own = take(nex);
    double* ex = world > 1 ? take(nex) : ex_own;
    double* SH[2] = {take(96), take(96)};
    double* rep_dev = take(32);
    {
        static bool fz_attr_set[64] = {};
        int dev_i = 0;
        ISLAM_HIP_CHECK(hipGetDevice(&dev_i));
        if (dev_i >= 0 && dev_i < 64 && !fz_attr_set[dev_i]) {
            ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)trial_elim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FZ_LDS_BYTES));
            fz_attr_set[dev_i] = true;
        }
    }
    static thread_local double* host_state = nullptr;
    if (!host_state) ISLAM_HIP_CHECK(hipHostMalloc((void**)&host_state, 32 * sizeof(double), hipHostMallocMapped | hipHostMallocPortable));
    double* report = nullptr;
    ISLAM_HIP_CHECK(hipHostGetDevicePointer((void**)&report, host_state, 0));
    volatile double* hs_all = host_state;
    hs_all[15] = 0.0;
    hs_all[31] = 0.0;
    // product rows of other ranks' segments read as zero; the own rows of the exchange buffer are rewritten by every solve
    if (world > 1)
        for (int l = 0; l < sp.nl; ++l) ISLAM_HIP_CHECK(hipMemsetAsync(w.lv[l].Dsep, 0, w.lv[l].prod_bytes, s));
    if (world > 1) ISLAM_HIP_CHECK(hipMemsetAsync(ex_own, 0, sizeof(double) * nex, s));
    {
        const int rc_init = enqueue_control_init(w, prm, s);
        if (rc_init != ISLAM_OK) return rc_init;
    }
    const TRParams tr{prm->high, prm->low, prm->up, prm->down, prm->factor, prm->rmin, prm->rmax, prm->reject,
                      prm->max_steps, prm->patience, prm->decreasing};
    const LinWeights W{prm->w[0], prm->w[1], prm->w[2], prm->w[3], prm->vmin, prm->vmax};
    double* LIN[2] = {w.lin, w.lin2};
    double* HD[2] = {w.Hd, w.Hd2};
    double* HO[2] = {w.Ho, w.Ho2};
    double* RH[2] = {w.rhs, w.rhs2};
    LevelBufs xb{};
    products_view(ex_own, Pxl, xb);
    auto level_out = [&](int l) {                              // products of level l: the exchange buffer at the exchange level
        LevelBufs ob = w.lv[l];
        if (l == xl) { ob.Dsep = xb.Dsep; ob.rsep = xb.rsep; ob.cL = xb.cL; ob.cR = xb.cR; ob.fill = xb.fill; ob.cgL = xb.cgL; ob.cgR = xb.cgR; }
        return ob;
    };
    int* const eflag_none = w.flags + 6;
    long long xbytes = 0;
    // levels 1 .. xl, the pack, the all-reduce, the decision and the down-sweep behind an eliminated level 0 of buffer pb
    // decision_only: an accepted trial would be the last optimizer step (StopOnPlateau's step limit) -- no solve follows it, only the
    // scalars of the message matter (the blocks in front of them are whatever the buffer holds, the same on every rank)
    auto enqueue_rest = [&](int pb, int mode, int damp_mode, bool with_trial, double seq, const Gate& gate, bool decision_only = false) -> int {
        for (int l = 1; l <= xl && !decision_only; ++l)
            launch_tw(level_src_from(w.lv[l - 1], sp.lv[l - 1].P), level_dst(level_out(l), nullptr), sp.lv[l].n, sp.lv[l].m, w.flags, R.seg0[l],
                      R.nseg[l], gate, s);
        PackArgs pa{};
        pa.f.n = xl;
        for (int l = 0; l < xl; ++l) { pa.f.cL[l] = w.lv[l].cL + (size_t)R.seg0[l] * 81; pa.f.cgL[l] = w.lv[l].cgL + (size_t)R.seg0[l] * 9; }
        if (has_left) { pa.f.Dsep = xb.Dsep + (size_t)(R.seg0[xl] - 1) * 81; pa.f.rsep = xb.rsep + (size_t)(R.seg0[xl] - 1) * 9; }
        pa.has_left = has_left; pa.has_right = has_right; pa.share = SH[pb]; pa.Hd_right = HD[pb] + (size_t)(N_eff - 1) * 81;
        pa.st = w.state; pa.tr = tr; pa.damp_mode = damp_mode; pa.part = with_trial ? w.part : (const double*)nullptr; pa.nwg = nwg;
        pa.msg = ex_own + 351 * (size_t)Pxl; pa.nmsg = nmsg; pa.rank = rank;
        // the verdict block: straight to the host's slot when no down-sweep follows, else to device memory -- the down-sweep forwards it
        double* const host_slot = report + 16 * ((long long)seq & 1);
        double* const dev_slot = rep_dev + 16 * ((long long)seq & 1);
        const bool forward = mode == 0 && !decision_only;
        DecideArgs da{};
        da.msg = ex + 351 * (size_t)Pxl; da.ex_Dsep = ex; da.st = w.state; da.tr = tr; da.report = forward ? dev_slot : host_slot; da.seq = seq;
        da.mode = mode; da.damp_mode = damp_mode; da.world = world; da.Pxl = Pxl; da.vmin = prm->vmin; da.vmax = prm->vmax;
        da.plain_report = forward ? 1 : 0;
        if (world > 1) {
            hipLaunchKernelGGL(shard_pack_kernel, dim3(1), dim3(128), 0, s, pa, gate);
            const int r = red.fn(red.self, ex_own, ex, nex, s);
            if (r != ISLAM_OK) return r;
            xbytes += 8LL * (long long)nex;
            hipLaunchKernelGGL(shard_decide_kernel, dim3(1), dim3(64), 0, s, da, gate);
        } else if (mode != 1) {          // (one rank, a solve without a trial: nothing to pack, nothing to decide)
            hipLaunchKernelGGL(shard_pack_decide_kernel, dim3(1), dim3(128), 0, s, pa, da, gate);
        }
        ISLAM_LAUNCH_CHECK();
        // (a trial that is not "accepted, continue, damping as speculated" bumps the epoch: the down-sweep turns into a no-op)
        if (decision_only) return ISLAM_OK;
        return shard_downsweep_planned(sp, R, w, ex, world, 0, w.dx, w.flags, gate, s, forward ? dev_slot : (const double*)nullptr, host_slot);
    };
    struct IterCfg { int pb; double *cur_n, *cur_v, *tri_n, *tri_v; };
    auto fused_args = [&](const IterCfg& c, bool first, double seq, int* eprev, bool trial_only = false) {
        FusedArgs fa{};
        fa.nodes = c.cur_n; fa.vels = c.cur_v; fa.dx = first ? (const double*)nullptr : w.dx; fa.poses = poses; fa.drots = drots; fa.dtrans = dtrans;
        fa.dvels = dvels; fa.dts = dts; fa.lin = first ? (const double*)nullptr : LIN[c.pb]; fa.N = N_eff; fa.nodes_t = c.tri_n; fa.vels_t = c.tri_v;
        fa.part = w.part; fa.st = w.state; fa.flags = w.flags; fa.ticket = nullptr; fa.tr = tr; fa.report = nullptr; fa.seq = seq; fa.W = W;
        const int ob = first ? c.pb : 1 - c.pb;
        fa.lin_o = LIN[ob]; fa.Hd_o = HD[ob]; fa.Ho_o = HO[ob]; fa.rhs_o = RH[ob];
        fa.dst = level_dst(level_out(0), w.dx);
        fa.m = m; fa.P = nseg; fa.nwg = nwg;
        fa.eflag = w.flags + 4 + (((long long)seq + 1) & 1);
        fa.eflag_prev = eprev;
        fa.Ms = M; fa.shard = 1; fa.seg_lo = seg_lo; fa.own_left = has_left ? 1 : 0; fa.share = SH[ob]; fa.open_right = has_right ? 1 : 0;
        fa.trial_only = trial_only ? 1 : 0;
        return fa;
    };
    IterCfg A{0, nodes, vels, w.nodes_t, w.vels_t};
    int steps = 0, trials = 0, status = ISLAM_OK;
    double loss = 0.0, damping = 1.0 / prm->radius, epoch = 1.0;
    auto run = [&]() -> int {
        int rc;
        {   // the first solve: linearisation at the initial iterate, its loss, elimination with the initial damping
            const Gate gate{w.state, epoch};
            const FusedArgs fa = fused_args(A, true, 0.0, eflag_none);
            hipLaunchKernelGGL(trial_elim_kernel, dim3(xcd_grid(nwg)), dim3(FZ_THREADS), FZ_LDS_BYTES, s, fa, gate);
            if ((rc = enqueue_rest(A.pb, 2, 2, true, 0.0, gate)) != ISLAM_OK) return rc;
        }
        // trial `seq` of iteration c: trial_elim_kernel (cur + dx -> tri, linearisation at tri, level 0 of solve seq+1) and the rest of
        // solve seq+1 around the all-reduce
        // steps_before: optimizer steps finished when this trial is evaluated
        auto enqueue_trial = [&](const IterCfg& c, double seq, double ep, bool prev_fused, int steps_before) -> int {
            const Gate gate{w.state, ep};
            const bool last = steps_before + 1 >= prm->max_steps;      // nothing can follow an accepted trial: decision only
            const FusedArgs fa = fused_args(c, false, seq, prev_fused ? w.flags + 4 + ((long long)seq & 1) : eflag_none, last);
            hipLaunchKernelGGL(trial_elim_kernel, dim3(xcd_grid(nwg)), dim3(FZ_THREADS), FZ_LDS_BYTES, s, fa, gate);
            return enqueue_rest(1 - c.pb, 0, 0, true, seq, gate, last);
        };
        if ((rc = enqueue_trial(A, 1.0, epoch, true, 0)) != ISLAM_OK) return rc;
        for (;;) {
            const double seq = (double)(trials + 1);
            // run ahead (the verdict of a trial is written BEHIND the all-reduce, too late to launch the next trial on time): trial
            // seq+1 under the assumption "accepted, continue, damping as speculated"; any other verdict bumps the device epoch and the
            // chain -- its collective included, on unchanged buffers, the same on every rank -- runs as no-ops
            const IterCfg B{1 - A.pb, A.tri_n, A.tri_v, A.cur_n, A.cur_v};
            if (steps + 1 < prm->max_steps && (rc = enqueue_trial(B, seq + 1.0, epoch, true, steps + 1)) != ISLAM_OK) return rc;
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
            if (verdict == 0) { A = B; continue; }              // B's trial is the one in flight
            epoch += 1.0;
            if (verdict == 2) { A = B; break; }
            if (verdict == 4) { status = ISLAM_ENOTPD; break; }
            // the speculative elimination is void: the next solve runs on the launched level-0 kernel from the linearisation in global
            // memory (undamped: the damping list of the state), with its own all-reduce of the interface blocks
            ISLAM_HIP_CHECK(hipMemsetAsync(w.flags + 4 + (((long long)seq + 1) & 1), 0, sizeof(int), s));
            if (verdict == 5) A = B;
            if (verdict == 3) status = ISLAM_ENOTPD;
            {
                const Gate gate{w.state, epoch};
                LevelSrc src{};
                src.level0 = 1; src.Hd = HD[A.pb]; src.Ho = HO[A.pb]; src.rhs0 = RH[A.pb]; src.state = w.state; src.hist = 1;
                launch_tw(src, level_dst(level_out(0), nullptr), N_eff, m, w.flags, seg_lo, nseg, gate, s);
                if ((rc = enqueue_rest(A.pb, 1, 1, false, seq, gate)) != ISLAM_OK) return rc;
            }
            if ((rc = enqueue_trial(A, seq + 1.0, epoch, false, steps)) != ISLAM_OK) return rc;
        }
        return ISLAM_OK;
    };
    const int rc = run();
    if (rc != ISLAM_OK) {
        hipLaunchKernelGGL(shard_close_gate_kernel, dim3(1), dim3(64), 0, s, w.state);
        (void)hipStreamSynchronize(s);
        (void)hipGetLastError();
        return rc;
    }
    res->steps = steps; res->trials = trials; res->status = status; res->loss = loss; res->damping = damping;
    if (exchanged_bytes) *exchanged_bytes = xbytes;
    *out_nodes = A.cur_n; *out_vels = A.cur_v;
    *own0 = has_left ? first_node : 0;
    *own1 = N_eff;
    return ISLAM_OK;
}

int trial_gated(const double* nodes, const double* vels, const double* dx, const double* poses, const double* drots,
                const double* dtrans, const double* dvels, const double* dts, const double* lin, int lin_stride, int M,
                double* nodes_t, double* vels_t, double* part, const double* red_lin, const double* red_trial,
                const islam_pvgo_reproj* reproj, int link0, Gate gate, hipStream_t s) {
    if (M < 1) return fail(ISLAM_EARG, "islam_pvgo_trial: M=%d < 1", M);
    ReprojDev rp{};
    if (reproj) {
        const int rc = reproj_dev_local(reproj, link0, rp);
        if (rc != ISLAM_OK) return rc;
    }
    hipLaunchKernelGGL(trial_kernel, dim3(xcd_grid((M + 63) / 64)), dim3(64), 0, s, nodes, vels, dx, poses, drots, dtrans, dvels, dts, lin,
                       M, nodes_t, vels_t, part, (double*)nullptr, (int*)nullptr, (unsigned*)nullptr, TRParams{}, (double*)nullptr, 0.0,
                       reproj ? red_lin : (const double*)nullptr, reproj ? red_trial : (const double*)nullptr, rp, lin_stride, gate);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int reproj_reduce_gated(const double* nodes, const double* dx, int M, const islam_pvgo_reproj* reproj, int link0, double* red,
                        Gate gate, hipStream_t s) {
    if (M < 1 || !reproj) return fail(ISLAM_EARG, "pvgo reproj reduce: M=%d or null reproj", M);
    ReprojDev rp{};
    const int rc = reproj_dev_local(reproj, link0, rp);
    if (rc != ISLAM_OK) return rc;
    enqueue_reproj_reduce(nodes, dx, M, rp, red, s, gate);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int linbuild_gated(const double* nodes, const double* vels, const double* poses, const double* drots, const double* dtrans,
                   const double* dvels, const double* dts, int N, const islam_pvgo_params* prm, double* lin, double* loss_part,
                   double* Hd, double* Ho, double* rhs, const double* red, const islam_pvgo_reproj* reproj, int link0, Gate gate,
                   hipStream_t s) {
    if (N < 2 || !prm) return fail(ISLAM_EARG, "pvgo linbuild: N=%d", N);
    int rc = ensure_linbuild_lds();
    if (rc != ISLAM_OK) return rc;
    ReprojDev rp{};
    if (reproj && (rc = reproj_dev_local(reproj, link0, rp)) != ISLAM_OK) return rc;
    const LinWeights W{prm->w[0], prm->w[1], prm->w[2], prm->w[3], prm->vmin, prm->vmax};
    const int nlb = (N + LB_NODES - 1) / LB_NODES;
    hipLaunchKernelGGL(linbuild_kernel, dim3(xcd_grid(nlb)), dim3(LB_THREADS), LB_DYN_BYTES, s, nodes, vels, poses, drots, dtrans, dvels, dts,
                       N, W, lin, loss_part, Hd, Ho, rhs, reproj ? red : (const double*)nullptr, rp, gate);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

}  // namespace islam

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

static int run_chain_impl(double* nodes, double* vels, const double* poses, const double* drots, const double* dtrans,
                          const double* dvels, const double* dts, int N, const islam_pvgo_params* prm,
                          const islam_pvgo_reproj* reproj, const ReprojDev& rp, Workspace& w, hipStream_t s,
                          islam_pvgo_result* result, double* trace, int trace_cap) {
    const int M = N - 1;
    // status blocks in pinned, device-visible host memory: the deciding wave of trial_lin_kernel writes one per trial
    // (two slots, alternating with the trial number), the host polls its sequence number (no stream synchronisation,
    // no copy on the critical path)
    static thread_local double* host_state = nullptr;
    if (!host_state) ISLAM_HIP_CHECK(hipHostMalloc((void**)&host_state, 32 * sizeof(double), hipHostMallocMapped | hipHostMallocPortable));   // portable: one buffer per thread serves calls on any device
    double* report = nullptr;
    ISLAM_HIP_CHECK(hipHostGetDevicePointer((void**)&report, host_state, 0));
    volatile double* hs_all = host_state;
    hs_all[15] = 0.0;
    hs_all[31] = 0.0;

    const double damping0 = 1.0 / prm->radius;               // TrustRegion: damping = 1/radius
    // device state and flags (flags[0] solver error, flags[2] ticket) initialised by a one-wave kernel: a host->device copy of a
    // stack array stalls the host for a staging round trip at the start of every run_pvgo
    {
        const int rc_init = enqueue_control_init(w, prm, s);
        if (rc_init != ISLAM_OK) return rc_init;
    }
    unsigned* ticket = reinterpret_cast<unsigned*>(w.flags + 2);
    TRParams tr{prm->high, prm->low, prm->up, prm->down, prm->factor, prm->rmin, prm->rmax, prm->reject,
                prm->max_steps, prm->patience, prm->decreasing};

    // Two linearisation buffers: trial_lin_kernel linearises at the trial point into the other buffer (the next
    // optimizer.step() if the trial is accepted -- the common case); on a reject the old buffer (with its cumulatively
    // damped diagonal) is simply kept.
    double* LIN[2] = {w.lin, w.lin2};
    double* HD[2] = {w.Hd, w.Hd2};
    double* HO[2] = {w.Ho, w.Ho2};
    double* RH[2] = {w.rhs, w.rhs2};
    double* RED[2] = {w.red, w.red2};
    const int nlb = (N + LB_NODES - 1) / LB_NODES;
    {
        const int rc_lds = ensure_linbuild_lds();
        if (rc_lds != ISLAM_OK) return rc_lds;
    }
    const LinWeights W{prm->w[0], prm->w[1], prm->w[2], prm->w[3], prm->vmin, prm->vmax};
    // red_ready: RED[b] already holds the reduction at (xn)
    auto enqueue_linbuild = [&](const double* xn, const double* xv, int b, bool red_ready) {
        if (reproj && !red_ready) enqueue_reproj_reduce(xn, nullptr, M, rp, RED[b], s);
        hipLaunchKernelGGL(linbuild_kernel, dim3(xcd_grid(nlb)), dim3(LB_THREADS), LB_DYN_BYTES, s, xn, xv, poses, drots, dtrans, dvels,
                           dts, N, W, LIN[b], w.loss_part, HD[b], HO[b], RH[b], reproj ? RED[b] : (const double*)nullptr, rp,
                           Gate{nullptr, 0.0});
    };
    // one pass of PyPose's inner `while self.last <= self.loss`: damped solve on buffer pb, then trial + linearisation at
    // the trial point into buffer 1-pb; every kernel is gated on `epoch`
    struct IterCfg { int pb; double *cur_n, *cur_v, *tri_n, *tri_v; };
    auto enqueue_iter = [&](const IterCfg& c, double seq, double epoch) -> int {
        const Gate gate{w.state, epoch};
        int rc = enqueue_solve(w, HD[c.pb], HO[c.pb], RH[c.pb], w.state, 0.0, N, prm->seg_len, w.dx, s, nullptr, nullptr, gate);
        if (rc != ISLAM_OK) return rc;
        // reprojection factor at the trial point Exp(dx)*cur: its r^T r joins the trial loss, and it IS the reduction of
        // the next linearisation if the trial is accepted
        if (reproj) enqueue_reproj_reduce(c.cur_n, w.dx, M, rp, RED[1 - c.pb], s, gate);
        double* rep_slot = report + 16 * ((long long)seq & 1);
        hipLaunchKernelGGL(trial_lin_kernel, dim3(xcd_grid(nlb) + 1), dim3(LB_THREADS), LB_DYN_BYTES, s, c.cur_n, c.cur_v, w.dx, poses, drots,
                           dtrans, dvels, dts, LIN[c.pb], N, c.tri_n, c.tri_v, w.part, w.state, w.flags, ticket, tr, rep_slot, seq,
                           reproj ? RED[c.pb] : (const double*)nullptr, reproj ? RED[1 - c.pb] : (const double*)nullptr, rp, W,
                           LIN[1 - c.pb], HD[1 - c.pb], HO[1 - c.pb], RH[1 - c.pb], gate);
        ISLAM_LAUNCH_CHECK();
        return ISLAM_OK;
    };

    IterCfg A{0, nodes, vels, w.nodes_t, w.vels_t};       // the iteration whose verdict is awaited
    int steps = 0, trials = 0, status = ISLAM_OK;
    double loss = 0.0, damping = damping0;
    double epoch = 1.0;
    // ---- the fused loop (default): trial t, the linearisation at its trial point and the level-0 elimination of solve t+1 in
    // ONE launch (trial_elim_kernel), under a speculated damping the deciding workgroup validates.  Plans it does not cover
    // (one-sided levels, segments longer than FZ_MAXM, a single level), the reprojection factor and ISLAM_PVGO_NO_FUSE=1 take
    // the launch-per-stage loop below.
    SolvePlan sp;
    plan_levels(N, prm->seg_len, sp, solve_twisted());
    const bool no_fuse = [] { const char* e = std::getenv("ISLAM_PVGO_NO_FUSE"); return e && e[0] == '1'; }();      // (read per call: A/B tests)
    // (one workgroup of FZ_S segments per CU: the whole level must be resident at once)
    // (the deciding workgroup is one more block with the same LDS footprint: it is dispatched to XCD 0, which must keep a CU free
    // for it -- otherwise it starts when the first workgroup exits and the launch ends ~4 us late)
    static const int fz_spare = [] { const char* e = std::getenv("ISLAM_FZ_SPARE"); return e ? std::atoi(e) : 16; }();      // (8 / 16 / 47 spare CUs: 63.3 / 62.9 / 62.9 us per LM iteration)
    const int fz_nwg = std::min(sp.lv[0].P, std::max(device_cus() - fz_spare, 1));
    // (small graphs -- the reference's own per-batch problem is 9 nodes, run_kitti.sh -- stay on the launch-per-stage loop: its launches
    // are cheaper than the fused kernel's fixed cost and a rejected trial costs no mis-speculated chain.  Measured per run_pvgo, fused /
    // launch-per-stage: N = 9 (18 trials) 1059 / 723 us, N = 65 206 / 190 us, N = 129 203 / 236 us, N = 513 443 / 508 us.)
    const bool fused = !no_fuse && !reproj && N > 96 && sp.twisted && sp.nl >= 2 && sp.top == sp.nl - 1 && sp.lv[0].m <= FZ_MAXM &&
                       prm->reject < STATE_DOUBLES - STATE_HIST - 1 && (sp.lv[0].P + fz_nwg - 1) / fz_nwg <= FZ_S;
    if (fused) {
        static bool fz_attr_set[64] = {};                        // per device: the attribute lives in the device's code object
        int dev_i = 0;
        ISLAM_HIP_CHECK(hipGetDevice(&dev_i));
        if (dev_i >= 0 && dev_i < 64 && !fz_attr_set[dev_i]) {
            ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)trial_elim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FZ_LDS_BYTES));
            fz_attr_set[dev_i] = true;
        }
        // every solve keeps the stored diagonal undamped and applies the damping history of the current linearisation (LevelSrc::hist)
        auto enqueue_solve_hist = [&](int pb, double ep) -> int {
            LevelSrc src{};
            src.level0 = 1; src.Hd = HD[pb]; src.Ho = HO[pb]; src.rhs0 = RH[pb]; src.state = w.state; src.hist = 1;
            return enqueue_levels(w, sp, 0, src, nullptr, w.dx, w.flags, s, nullptr, nullptr, Gate{w.state, ep});
        };
        int* const eflag_none = w.flags + 6;                 // a word nobody sets
        bool begin_pending = true;                           // the initial loss has not been summed into the state yet
        // evaluates trial `seq` of iteration c (cur + dx -> tri); more: also eliminates level 0 of solve seq+1 and enqueues its upper
        // levels + down-sweep (-> dx).  prev_fused: level 0 of solve `seq` ran inside the previous trial_elim_kernel.
        auto enqueue_trial = [&](const IterCfg& c, double seq, double ep, bool more, bool prev_fused) -> int {
            const Gate gate{w.state, ep};
            double* rep_slot = report + 16 * ((long long)seq & 1);
            int* eprev = prev_fused ? w.flags + 4 + ((long long)seq & 1) : eflag_none;
            if (!more && !begin_pending) {   // nothing follows an accepted trial: the fused kernel's trial-only mode (no node blocks, no
                FusedArgs fa{};              // elimination; 7 us against trial_lin_kernel's 10.6)
                fa.nodes = c.cur_n; fa.vels = c.cur_v; fa.dx = w.dx; fa.poses = poses; fa.drots = drots; fa.dtrans = dtrans; fa.dvels = dvels;
                fa.dts = dts; fa.lin = LIN[c.pb]; fa.N = N; fa.nodes_t = c.tri_n; fa.vels_t = c.tri_v; fa.part = w.part; fa.st = w.state;
                fa.flags = w.flags; fa.ticket = ticket; fa.tr = tr; fa.report = rep_slot; fa.seq = seq; fa.W = W;
                fa.lin_o = LIN[1 - c.pb]; fa.Hd_o = HD[1 - c.pb]; fa.Ho_o = HO[1 - c.pb]; fa.rhs_o = RH[1 - c.pb];
                fa.dst = level_dst(w.lv[0], w.dx);
                fa.m = sp.lv[0].m; fa.P = sp.lv[0].P; fa.nwg = fz_nwg;
                fa.eflag = eflag_none;
                fa.eflag_prev = eprev;
                fa.Ms = M;
                fa.trial_only = 1;
                hipLaunchKernelGGL(trial_elim_kernel, dim3(xcd_grid(fz_nwg) + 1), dim3(FZ_THREADS), FZ_LDS_BYTES, s, fa, gate);
                ISLAM_LAUNCH_CHECK();
                return ISLAM_OK;
            }
            if (!more) {                     // (the run's first trial is also its last: the initial loss still has to be summed)
                if (begin_pending) {
                    hipLaunchKernelGGL(control_begin_kernel, dim3(1), dim3(64), 0, s, w.loss_part, nlb, w.state, w.flags);
                    begin_pending = false;
                }
                hipLaunchKernelGGL(trial_lin_kernel, dim3(xcd_grid(nlb) + 1), dim3(LB_THREADS), LB_DYN_BYTES, s, c.cur_n, c.cur_v, w.dx, poses,
                                   drots, dtrans, dvels, dts, LIN[c.pb], N, c.tri_n, c.tri_v, w.part, w.state, w.flags, ticket, tr, rep_slot,
                                   seq, (const double*)nullptr, (const double*)nullptr, rp, W, (double*)nullptr, (double*)nullptr,
                                   (double*)nullptr, (double*)nullptr, gate, eprev);
                ISLAM_LAUNCH_CHECK();
                return ISLAM_OK;
            }
            FusedArgs fa{};
            fa.nodes = c.cur_n; fa.vels = c.cur_v; fa.dx = w.dx; fa.poses = poses; fa.drots = drots; fa.dtrans = dtrans; fa.dvels = dvels;
            fa.dts = dts; fa.lin = LIN[c.pb]; fa.N = N; fa.nodes_t = c.tri_n; fa.vels_t = c.tri_v; fa.part = w.part; fa.st = w.state;
            fa.flags = w.flags; fa.ticket = ticket; fa.tr = tr; fa.report = rep_slot; fa.seq = seq; fa.W = W;
            fa.lin_o = LIN[1 - c.pb]; fa.Hd_o = HD[1 - c.pb]; fa.Ho_o = HO[1 - c.pb]; fa.rhs_o = RH[1 - c.pb];
            fa.dst = level_dst(w.lv[0], w.dx);
            fa.m = sp.lv[0].m; fa.P = sp.lv[0].P; fa.nwg = fz_nwg;
            fa.eflag = w.flags + 4 + (((long long)seq + 1) & 1);
            fa.eflag_prev = eprev;
            fa.Ms = M;
            if (begin_pending) { fa.loss_part0 = w.loss_part; fa.nlb0 = nlb; begin_pending = false; }
            hipLaunchKernelGGL(trial_elim_kernel, dim3(xcd_grid(fz_nwg) + 1), dim3(FZ_THREADS), FZ_LDS_BYTES, s, fa, gate);
            LevelSrc none{};
            none.level0 = 1;
            return enqueue_levels(w, sp, 0, none, nullptr, w.dx, w.flags, s, nullptr, nullptr, gate, true);
        };
        // the first solve: the same kernel in its `first` mode (dx = nullptr) linearises at the initial iterate, sums the initial loss
        // and eliminates level 0 with the initial damping -- linbuild_kernel + the launched level-0 kernel only on the fallback paths
        int rc;
        {
            const Gate gate{w.state, epoch};
            FusedArgs fa{};
            fa.nodes = A.cur_n; fa.vels = A.cur_v; fa.dx = nullptr; fa.poses = poses; fa.drots = drots; fa.dtrans = dtrans; fa.dvels = dvels;
            fa.dts = dts; fa.lin = nullptr; fa.N = N; fa.nodes_t = A.tri_n; fa.vels_t = A.tri_v; fa.part = w.part; fa.st = w.state;
            fa.flags = w.flags; fa.ticket = ticket; fa.tr = tr; fa.report = nullptr; fa.seq = 0.0; fa.W = W;
            fa.lin_o = LIN[A.pb]; fa.Hd_o = HD[A.pb]; fa.Ho_o = HO[A.pb]; fa.rhs_o = RH[A.pb];
            fa.dst = level_dst(w.lv[0], w.dx);
            fa.m = sp.lv[0].m; fa.P = sp.lv[0].P; fa.nwg = fz_nwg;
            fa.eflag = w.flags + 4 + 1;                          // solve 1
            fa.eflag_prev = eflag_none;
            fa.Ms = M;
            hipLaunchKernelGGL(trial_elim_kernel, dim3(xcd_grid(fz_nwg) + 1), dim3(FZ_THREADS), FZ_LDS_BYTES, s, fa, gate);
            LevelSrc none{};
            none.level0 = 1;
            rc = enqueue_levels(w, sp, 0, none, nullptr, w.dx, w.flags, s, nullptr, nullptr, gate, true);
            begin_pending = false;
        }
        if (rc != ISLAM_OK) return rc;
        bool prev_fused = true;
        for (;;) {
            const double seq = (double)(trials + 1);
            const IterCfg B{1 - A.pb, A.tri_n, A.tri_v, A.cur_n, A.cur_v};
            // (an accepted trial that would be the last optimizer step anyway -- StopOnPlateau's step limit -- needs no next solve)
            const bool more = steps + 1 < prm->max_steps;
            rc = enqueue_trial(A, seq, epoch, more, prev_fused);
            if (rc != ISLAM_OK) return rc;
            volatile double* hs = hs_all + 16 * ((long long)seq & 1);
            {
                unsigned long spins = 0;
                while (hs[15] != seq) {
                    if (++spins > 400000000ul) {
                        ISLAM_HIP_CHECK(hipStreamSynchronize(s));
                        if (hs[15] != seq) return fail(ISLAM_EHIP, "islam_pvgo_run_chain: no status from the device (trial %d)", trials + 1);
                    }
                }
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
            }
            ++trials;
            const int verdict = (int)hs[12];
            if (verdict == 9) return fail(ISLAM_EHIP, "islam_pvgo_run_chain: the deciding workgroup of trial %d gave up waiting for the level's workgroups", trials);
            damping = hs[2];
            loss = hs[0];
            steps = (int)hs[13];
            if (trace && trials <= trace_cap && verdict != 3 && verdict != 4) {
                trace[3 * (trials - 1)] = hs[6];
                trace[3 * (trials - 1) + 1] = damping;
                trace[3 * (trials - 1) + 2] = (verdict == 1) ? 0.0 : 1.0;
            }
            if (verdict == 0) {               // accepted, the speculated damping was right: solve seq+1 is already running
                A = B;
                prev_fused = true;
                continue;
            }
            epoch += 1.0;                     // any other verdict bumped the device epoch: the launches queued behind are no-ops
            if (verdict == 2) { A = B; break; }
            if (verdict == 4) { status = ISLAM_ENOTPD; break; }
            // the speculative level-0 elimination (if there was one) is void: clear its error word; the next solve runs on the
            // launched kernels from the linearisation in global memory
            if (more) ISLAM_HIP_CHECK(hipMemsetAsync(w.flags + 4 + (((long long)seq + 1) & 1), 0, sizeof(int), s));
            if (verdict == 5) A = B;          // accepted with another damping: the trial point's linearisation is in the other buffers
            if (verdict == 3) status = ISLAM_ENOTPD;      // "Linear solver failed. Breaking optimization step...": same iterate, same
                                                          // (undamped) linearisation, StopOnPlateau's plateau counter ends the loop
            rc = enqueue_solve_hist(A.pb, epoch);
            if (rc != ISLAM_OK) return rc;
            prev_fused = false;
        }
        if (A.cur_n != nodes) {
            ISLAM_HIP_CHECK(hipMemcpyAsync(nodes, A.cur_n, (size_t)N * 7 * sizeof(double), hipMemcpyDeviceToDevice, s));
            ISLAM_HIP_CHECK(hipMemcpyAsync(vels, A.cur_v, (size_t)N * 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        result->steps = steps;
        result->trials = trials;
        result->status = status;
        result->loss = loss;
        result->damping = damping;
        return ISLAM_OK;
    }
    enqueue_linbuild(A.cur_n, A.cur_v, A.pb, false);
    hipLaunchKernelGGL(control_begin_kernel, dim3(1), dim3(64), 0, s, w.loss_part, nlb, w.state, w.flags);
    // ---- small graphs (one segment, one block of links: the reference's own per-batch window of 9 nodes): the whole loop in ONE launch
    const bool no_small = [] { const char* e = std::getenv("ISLAM_PVGO_NO_SMALL"); return e && e[0] == '1'; }();      // (read per call: A/B tests)
    // (one wave eliminates the window's nodes one after the other, ~2 us each: beyond a couple of dozen nodes the level tree of the
    // launch-per-stage loop is faster -- N = 65 takes 190 us per run there)
    constexpr int SMALL_MAX_N = 16;
    if (!no_small && !reproj && N <= SMALL_MAX_N) {
        constexpr int SMALL_LDS = LB_DYN_BYTES + LDS_PER_WAVE * (int)sizeof(double);
        static bool small_attr_set[64] = {};
        int dev_i = 0;
        ISLAM_HIP_CHECK(hipGetDevice(&dev_i));
        if (dev_i >= 0 && dev_i < 64 && !small_attr_set[dev_i]) {
            ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)small_lm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMALL_LDS));
            small_attr_set[dev_i] = true;
        }
        static thread_local double* host_trace = nullptr;       // pinned rows for the optional trace (3 per trial)
        constexpr int TRACE_ROWS = 1024;
        double* trace_dev = nullptr;
        if (trace && trace_cap > 0) {
            if (!host_trace) ISLAM_HIP_CHECK(hipHostMalloc((void**)&host_trace, 3 * TRACE_ROWS * sizeof(double), hipHostMallocMapped | hipHostMallocPortable));
            ISLAM_HIP_CHECK(hipHostGetDevicePointer((void**)&trace_dev, host_trace, 0));
        }
        SmallArgs sa{};
        sa.nodes = nodes; sa.vels = vels; sa.poses = poses; sa.drots = drots; sa.dtrans = dtrans; sa.dvels = dvels; sa.dts = dts; sa.N = N;
        sa.nodes_t = w.nodes_t; sa.vels_t = w.vels_t; sa.dx = w.dx;
        for (int i = 0; i < 2; ++i) { sa.LIN[i] = LIN[i]; sa.HD[i] = HD[i]; sa.HO[i] = HO[i]; sa.RH[i] = RH[i]; }
        sa.loss_part = w.loss_part; sa.st = w.state; sa.flags = w.flags; sa.tr = tr; sa.W = W;
        sa.dst = level_dst(w.lv[0], w.dx);
        sa.report = report; sa.trace = trace_dev; sa.trace_cap = std::min(trace_cap, TRACE_ROWS); sa.marker = 7.0;
        hipLaunchKernelGGL(small_lm_kernel, dim3(1), dim3(LB_THREADS), SMALL_LDS, s, sa);
        ISLAM_LAUNCH_CHECK();
        {
            unsigned long spins = 0;
            while (hs_all[15] != sa.marker) {
                if (++spins > 400000000ul) {
                    ISLAM_HIP_CHECK(hipStreamSynchronize(s));
                    if (hs_all[15] != sa.marker) return fail(ISLAM_EHIP, "islam_pvgo_run_chain: no status from the device (small-graph loop)");
                }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        result->loss = hs_all[0];
        result->damping = hs_all[2];
        result->status = (int)hs_all[10];
        result->trials = (int)hs_all[11];
        result->steps = (int)hs_all[13];
        if (trace && trace_cap > 0) {
            const int nt = std::min(result->trials, sa.trace_cap);
            for (int i = 0; i < 3 * nt; ++i) trace[i] = host_trace[i];
        }
        return ISLAM_OK;
    }
    int rc = enqueue_iter(A, 1.0, epoch);
    if (rc != ISLAM_OK) return rc;
    for (;;) {
        const double seq = (double)(trials + 1);
        // run ahead: the next iteration under the assumption "trial accepted, loop continues" -- unless an accepted trial
        // would be the last optimizer step anyway (StopOnPlateau's step limit): nothing can follow it
        const IterCfg B{1 - A.pb, A.tri_n, A.tri_v, A.cur_n, A.cur_v};
        if (steps + 1 < prm->max_steps) {
            rc = enqueue_iter(B, seq + 1.0, epoch);
            if (rc != ISLAM_OK) return rc;
        }
        // wait for the verdict (poll the pinned status block; fall back to a stream sync after ~2 s)
        volatile double* hs = hs_all + 16 * ((long long)seq & 1);
        {
            unsigned long spins = 0;
            while (hs[15] != seq) {
                if (++spins > 400000000ul) {
                    ISLAM_HIP_CHECK(hipStreamSynchronize(s));
                    if (hs[15] != seq) return fail(ISLAM_EHIP, "islam_pvgo_run_chain: no status from the device (trial %d)", trials + 1);
                }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        ++trials;
        const int verdict = (int)hs[12];
        damping = hs[2];
        loss = hs[0];
        steps = (int)hs[13];
        if (trace && trials <= trace_cap && verdict < 3) {
            trace[3 * (trials - 1)] = hs[6];
            trace[3 * (trials - 1) + 1] = damping;
            trace[3 * (trials - 1) + 2] = (verdict == 1) ? 0.0 : 1.0;
        }
        if (verdict == 0) {               // accepted, continue: B is the iteration now in flight
            A = B;
            continue;
        }
        epoch += 1.0;                     // any other verdict bumped the device epoch: B's kernels are no-ops
        if (verdict == 1) {               // rejected: same iterate, same (cumulatively damped) linearisation
            rc = enqueue_iter(A, seq + 1.0, epoch);
            if (rc != ISLAM_OK) return rc;
            continue;
        }
        if (verdict == 2) {               // accepted, StopOnPlateau says stop
            A = B;
            break;
        }
        status = ISLAM_ENOTPD;            // "Linear solver failed. Breaking optimization step..."
        if (verdict == 4) break;
        // PyPose keeps looping through the scheduler (the plateau counter stops it): same iterate, new linearisation
        enqueue_linbuild(A.cur_n, A.cur_v, A.pb, true);
        rc = enqueue_iter(A, seq + 1.0, epoch);
        if (rc != ISLAM_OK) return rc;
    }
    double* cur_n = A.cur_n;
    double* cur_v = A.cur_v;
    if (cur_n != nodes) {
        ISLAM_HIP_CHECK(hipMemcpyAsync(nodes, cur_n, (size_t)N * 7 * sizeof(double), hipMemcpyDeviceToDevice, s));
        ISLAM_HIP_CHECK(hipMemcpyAsync(vels, cur_v, (size_t)N * 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    result->steps = steps;
    result->trials = trials;
    result->status = status;
    result->loss = loss;
    result->damping = damping;
    return ISLAM_OK;
}

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
