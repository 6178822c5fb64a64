// pvgo_lm_kernels.inl -- part of the pvgo.hip translation unit (textually included there; not compiled on its own).
// device side of the LM loop: trial step, the fused trial + linearisation + level-0 elimination kernel, the one-launch loop for small graphs (pp.optim.LM.step, TrustRegion, StopOnPlateau)
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
    // two-level solve (m0 > 0): the window as two segments of m0 nodes around a separator, eliminated by two wavefronts side by side,
    // then the 1- or 2-node root (bt_top_kernel's scheme inside this workgroup): 4 + 1 dependent node steps each way instead of 9
    LevelSrc src1;
    LevelDst dst1;
    int m0, P0, n1;
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
        if (a.m0 > 0) {
            if (wave < a.P0) {
                LevelSrc src{};
                src.level0 = 1; src.Hd = a.HD[pb]; src.Ho = a.HO[pb]; src.rhs0 = a.RH[pb]; src.state = a.st; src.damping_override = 0.0;
                eliminate_segment(src, a.dst, N, a.m0, wave, a.flags, lane, lds_solve + wave * LDS_PER_WAVE);
            }
            __syncthreads();                                     // the two segments' products are visible to the workgroup
            if (wave == 0) {
                eliminate_segment(a.src1, a.dst1, a.n1, a.n1, 0, a.flags, lane, lds_solve);
                double xn[9], xL[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) { xn[q] = 0.0; xL[q] = 0.0; }
                backsub_segment(a.dst1.fac, a.dst1.inv, a.dst1.x, 0, a.n1, lane, xn, xL);
            }
            __syncthreads();
            if (wave < a.P0) backsub_level_segment(a.dst.fac, a.dst.inv, a.dst1.x, a.dx, N, a.m0, wave, lane);
        } else if (wave == 0) {
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

