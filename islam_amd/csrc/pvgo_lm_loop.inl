// pvgo_lm_loop.inl -- part of the pvgo.hip translation unit (textually included there; not compiled on its own).
// run_chain_impl: the host loop of islam_pvgo_run_chain[_reproj] (reference pvgo.py:168-180), run-ahead gate, verdict polling
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
        constexpr int SMALL_LDS = LB_DYN_BYTES + 2 * LDS_PER_WAVE * (int)sizeof(double);
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
        {
            // two segments around a separator (ISLAM_SMALL_LM_ONE_SEGMENT=1: the one-segment sweep of rounds 3-5, A/B runs)
            const bool one = [] { const char* e = std::getenv("ISLAM_SMALL_LM_ONE_SEGMENT"); return e && e[0] == '1'; }();
            const int m0 = (!one && N >= 3) ? (N - 1) / 2 : 0;
            sa.m0 = m0;
            if (m0 > 0) {
                sa.P0 = (N + m0) / (m0 + 1);
                sa.n1 = N / (m0 + 1);
                sa.src1 = level_src_from(w.lv[0], sa.P0);
                sa.dst1 = level_dst(w.lv[1], w.lv[1].x);
            }
        }
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
