// pvgo_host.inl -- part of the pvgo.hip translation unit (textually included there; not compiled on its own).
// host side: level planner, workspace carving, kernel launchers
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
    // level 0 of a graph with >= 3072 segments (four rounds of resident workgroups) runs on the two-wave kernel: N = 300 007, 37 504
    // segments: 1266 -> 1205 us per LM iteration.  (ISLAM_PVGO_L0_TW2 = that threshold; 0: never)
    static const int tw2_from = [] { const char* e = std::getenv("ISLAM_PVGO_L0_TW2"); return e ? std::atoi(e) : 3072; }();
    if (src.level0 && tw2_from > 0 && nseg >= tw2_from) {
        hipLaunchKernelGGL(bt_eliminate_tw2_kernel, dim3(xcd_grid(nseg)), dim3(128), 0, s, src, dst, n, m, flags, seg0, nseg, gate);
        return;
    }
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

