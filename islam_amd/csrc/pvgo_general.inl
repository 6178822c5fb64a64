// pvgo_general.inl -- part of the pvgo.hip translation unit (textually included there; not compiled on its own).
// control kernels, retraction, arbitrary-topology assembly, vo_loss / imu_loss forward + backward, align_to (reference pvgo.py:67-119)
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

