// pvgo_linearize.inl -- part of the pvgo.hip translation unit (textually included there; not compiled on its own).
// link residuals, Jacobian blocks, normal equations, the sparse reprojection factor, linbuild_kernel (reference pvgo.py:26-64, dense_ba.py:276-305)
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

