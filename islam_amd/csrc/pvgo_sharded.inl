// pvgo_sharded.inl -- part of the pvgo.hip translation unit (textually included there; not compiled on its own).
// the sharded LM loop (one graph over several GPUs; driven by pvgo_dist.hip): pack / decide kernels, run_chain_sharded_fused
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
