// pvgo_solver.inl -- part of the pvgo.hip translation unit (textually included there; not compiled on its own).
// partitioned / twisted block-tridiagonal LDL^T: eliminate, top, back-substitution and down-sweep kernels (what ppos.Cholesky does on the dense matrix)
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

// Level 0 of a LARGE graph (far more segments than the chip holds at once, launch_tw): two wavefronts per segment, no helper -- a
// workgroup is 128 threads and 2/3 of the LDS, so half again as many segments are in flight per CU; what the helper took off the
// sweeping waves' critical path matters less than residency once a level runs in dozens of rounds.  Same factor layout.
__global__ __launch_bounds__(128, 3) void bt_eliminate_tw2_kernel(LevelSrc src, LevelDst dst, int n, int m, int* flags, int seg0, int nseg, Gate gate) {
    __shared__ __attribute__((aligned(16))) double lds[LDS_TWISTED];
    const int p = xcd_index(blockIdx.x, nseg);
    if (p < 0) return;
    eliminate_twisted<1>(src, dst, n, m, p + seg0, flags, threadIdx.x >> 6, threadIdx.x & 63, lds, gate);
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

