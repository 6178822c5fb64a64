// PWC-Net cost-volume kernels for gfx950: 81-channel local correlation (forward + both gradients)
// and the backward warp with validity mask.
//
// Replaces (reference file:line under /root/reference):
//   Network/PWC/correlation.py:8-33    kernel_Correlation_rearrange   (NCHW -> zero-padded NHWC copy in HBM)
//   Network/PWC/correlation.py:35-103  kernel_Correlation_updateOutput (1 block of 32 threads per pixel,
//                                       81 serial __syncthreads rounds)
//   Network/PWC/correlation.py:105-233 kernel_Correlation_updateGradFirst / updateGradSecond
//   Network/PWC/PWCNet.py:170-206      PWCDCNet.warp (CPU meshgrid + H2D + 2 grid_samples + masking)
//
// MI355X design: no padded copy ever touches HBM.  A workgroup owns a 32x4 pixel tile of one image;
// it streams the channels in chunks of 16 through LDS (f1 tile and the (4+8)x(32+8) halo tile of f2,
// zero filled outside the image, coalesced 128-byte row reads of the NCHW tensors).  The nine waves
// of the workgroup each own one vertical displacement dy; a lane owns two horizontally adjacent
// pixels and keeps their 2x9 horizontal displacements in registers, so every f2 value fetched from
// LDS (five 8-byte reads per channel) feeds up to four FMAs.  Writes are 128-byte row segments.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "common.h"

using namespace islam;

namespace {

constexpr int TW = 32, TH = 4;          // pixel tile
constexpr int CC = 16;                  // channels per LDS chunk
constexpr int F2W = TW + 8, F2H = TH + 8;
constexpr int F2RS = F2W;               // row stride (floats), even -> 8-byte aligned pairs

constexpr int NT = 576;                                  // 16 x 4 x 9 threads
constexpr int STG2 = (CC * F2H * F2W + NT - 1) / NT;     // staging registers per thread (f2 halo tile)
constexpr int STG1 = (CC * TH * TW + NT - 1) / NT;       // (f1 tile)

// One workgroup = one 32x4 pixel tile of one image and one channel slice [c_begin, c_end).  With a single slice the
// result (sum / C) goes straight to `out`; with several slices (small pyramid levels: too few tiles to fill 256 CUs)
// every slice writes its partial sum to `part[slice]` and corr81_reduce_kernel adds the slices in index order.
__global__ __launch_bounds__(NT) void corr81_fwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                         float* __restrict__ out, float* __restrict__ part, int B, int C,
                                                         int H, int W, int nslice, int cps, int otot, int ooff, float slope) {
    __shared__ __attribute__((aligned(16))) float s1[CC * TH * TW];
    __shared__ __attribute__((aligned(16))) float s2[CC * F2H * F2RS];
    const int tx = threadIdx.x, ty = threadIdx.y, dyi = threadIdx.z;      // 16 x 4 x 9
    const int tid = tx + 16 * ty + 64 * dyi;
    const int b = blockIdx.z / nslice, slice = blockIdx.z - b * nslice;
    const int c_begin = slice * cps, c_end = min(C, c_begin + cps);
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const size_t plane = (size_t)H * W;
    const float* f1b = f1 + (size_t)b * C * plane;
    const float* f2b = f2 + (size_t)b * C * plane;

    float acc0[9], acc1[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    // staging map, identical for every chunk: element q of this lane -> (channel in chunk, global offset, LDS slot)
    int g2[STG2], l2[STG2], c2[STG2], g1[STG1], c1[STG1];
#pragma unroll
    for (int q = 0; q < STG2; ++q) {
        const int i = tid + q * NT;
        const int c = i / (F2H * F2W), rem = i - c * (F2H * F2W);
        const int ly = rem / F2W, lx = rem - ly * F2W;
        const int gy = y0 - 4 + ly, gx = x0 - 4 + lx;
        const bool ok = i < CC * F2H * F2W && gy >= 0 && gy < H && gx >= 0 && gx < W;
        c2[q] = ok ? c : CC;                              // CC = never fetched
        g2[q] = ok ? gy * W + gx : 0;
        l2[q] = i < CC * F2H * F2W ? (c * F2H + ly) * F2RS + lx : -1;
    }
#pragma unroll
    for (int q = 0; q < STG1; ++q) {
        const int i = tid + q * NT;
        const int c = i / (TH * TW), rem = i - c * (TH * TW);
        const int ly = rem / TW, lx = rem - ly * TW;
        const int gy = y0 + ly, gx = x0 + lx;
        const bool ok = i < CC * TH * TW && gy < H && gx < W;
        c1[q] = ok ? c : CC;
        g1[q] = ok ? gy * W + gx : 0;
    }
    // software pipeline over channel chunks: the global loads of chunk k+1 are in flight (in registers) while chunk k
    // is consumed from LDS
    float r2[STG2], r1[STG1];
    auto fetch = [&](int cb, int nc) {
#pragma unroll
        for (int q = 0; q < STG2; ++q) r2[q] = c2[q] < nc ? f2b[(size_t)(cb + c2[q]) * plane + g2[q]] : 0.f;
#pragma unroll
        for (int q = 0; q < STG1; ++q) r1[q] = c1[q] < nc ? f1b[(size_t)(cb + c1[q]) * plane + g1[q]] : 0.f;
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < STG2; ++q)
            if (l2[q] >= 0) s2[l2[q]] = r2[q];
#pragma unroll
        for (int q = 0; q < STG1; ++q)
            if (tid + q * NT < CC * TH * TW) s1[tid + q * NT] = r1[q];
    };

    fetch(c_begin, min(CC, c_end - c_begin));
    for (int cb = c_begin; cb < c_end; cb += CC) {
        const int nc = min(CC, c_end - cb);
        __syncthreads();                         // previous chunk fully consumed
        commit();
        __syncthreads();
        if (cb + CC < c_end) fetch(cb + CC, min(CC, c_end - cb - CC));
        for (int c = 0; c < nc; ++c) {
            const float2 a = *reinterpret_cast<const float2*>(&s1[(c * TH + ty) * TW + 2 * tx]);
            const float* row = &s2[(c * F2H + ty + dyi) * F2RS + 2 * tx];
            float r[10];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const float2 v = *reinterpret_cast<const float2*>(row + 2 * q);
                r[2 * q] = v.x;
                r[2 * q + 1] = v.y;
            }
#pragma unroll
            for (int dx = 0; dx < 9; ++dx) {
                acc0[dx] = fmaf(a.x, r[dx], acc0[dx]);
                acc1[dx] = fmaf(a.y, r[dx + 1], acc1[dx]);
            }
        }
    }
    const int gy = y0 + ty, gx = x0 + 2 * tx;
    if (gy < H && gx < W) {
        const float sc = nslice == 1 ? 1.0f / (float)C : 1.0f;          // the mean over channels (correlation.py:97-99)
        // single slice: straight into channels [ooff, ooff + 81) of the (B, otot, H, W) destination, LeakyReLU(slope) applied (slope 1: none)
        float* ob = (nslice == 1 ? out + ((size_t)b * otot + ooff + (size_t)dyi * 9) * plane
                                 : part + (size_t)slice * B * 81 * plane + ((size_t)b * 81 + (size_t)dyi * 9) * plane) + (size_t)gy * W + gx;
        const float sl = nslice == 1 ? slope : 1.0f;
        auto act = [&](float v) { return v > 0.0f ? v : v * sl; };
        const bool pair = (gx + 1 < W) && ((((size_t)gy * W + gx) & 1) == 0) && ((plane & 1) == 0);   // 8-byte aligned pair
#pragma unroll
        for (int dx = 0; dx < 9; ++dx) {
            if (pair) {
                *reinterpret_cast<float2*>(ob + (size_t)dx * plane) = make_float2(act(acc0[dx] * sc), act(acc1[dx] * sc));
            } else {
                ob[(size_t)dx * plane] = act(acc0[dx] * sc);
                if (gx + 1 < W) ob[(size_t)dx * plane + 1] = act(acc1[dx] * sc);
            }
        }
    }
}

// Four pixels per lane (same arithmetic, same argument meaning as corr81_fwd_kernel above, which stays for W % 4 != 0).  The
// one-pixel-pair form issues per channel and lane six 8-byte LDS reads for 18 FMAs and stages its tiles with 4-byte global loads and
// LDS writes: it is bound by instruction issue (LDS + vector-memory requests), 72 us for the level-2 call (83 MB: 0.14 of the HBM
// roof).  Here a lane owns four horizontally adjacent pixels x nine horizontal displacements of ONE vertical displacement (36
// accumulators): per channel one 16-byte read of f1 and three of the f2 halo row feed 36 FMAs; tiles are staged with 16-byte global
// loads / LDS writes, results leave as 16-byte stores (128-byte row segments of a displacement plane).  Needs W % 4 == 0 and 16-byte
// aligned tensors.
//
// Work layout (round 6).  The unit of work is a SLOT: 8 lanes = one 32-pixel tile row of one vertical displacement.  A workgroup owns
// a 32 x 7 pixel tile = 9 x 7 = 63 slots on 8 waves (the 64th slot idles): two waves on every SIMD, and at <= 128 registers two such
// workgroups per CU.  Rounds 3-5 ran a 32 x 8 tile on NINE waves (one per vertical displacement, 72 slots): phase clocks
// (scripts/debug/corr_stamps.py) showed the multiply phase to be what the call costs -- with global loads AND stores compiled out
// the level-2 call still took 35 of its 36 us -- and one SIMD carrying three of the nine waves: the others waited 2-3 us per tile at
// the next barrier, and 560 tiles on 256 one-workgroup CUs meant three rounds for 2.2 tiles' worth of work.  Measured on one box,
// level 2 (B = 8, C = 32, 112 x 160): 35.5 us -> 30.4 (7 rows, 8 waves, one workgroup per CU) -> 24.9 (two per CU: no spills, see
// `fetch`) = 0.29 -> 0.42 of the HBM roof.  Measured and dropped on the way: requesting channel c + 1's LDS operands before channel c
// is multiplied (+16 registers; 2-3 us SLOWER in every configuration: the phase is bound by issue, not by LDS latency), storing a
// tile's results behind the next tile's first commit (slower), 3 x 128 / 3 x 256 workgroups (same / slower).  H = 7, 14, 28, 56, 112
// (every PWC level of a 448-row image) are whole tiles.
constexpr int TW4 = 32, TH4 = 7, F2W4 = TW4 + 8, F2H4 = TH4 + 8;
constexpr int NT4 = 512;                                         // 8 waves = 64 slots of 8 lanes; 63 of them = 9 displacements x 7 rows
constexpr int Q2 = (CC * F2H4 * (F2W4 / 4) + NT4 - 1) / NT4;   // float4 items of the f2 halo tile per thread (2400 / 512 -> 5)
constexpr int Q1 = (CC * TH4 * (TW4 / 4) + NT4 - 1) / NT4;     // float4 items of the f1 tile per thread (896 / 512 -> 2)

typedef float v4f __attribute__((ext_vector_type(4)));

#ifdef ISLAM_CORR_STAMPS               // scripts/debug/corr_stamps.sh: event clocks of workgroup 0's wave 0 (never in the product build)
}  // namespace
__device__ long long islam_corr_stamps_buf[64];
__device__ int islam_corr_dbg;            // 1: no stores, 2: no global loads (what the other half costs on its own)
namespace {
#define CSTAMP() do { __builtin_amdgcn_sched_barrier(0); if (stamp && st_n < 62) st_ev[st_n++] = wall_clock64() - st_t0; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CSTAMP() do { } while (0)
#endif

// Persistent form: the launch has two workgroups per CU, workgroup i walks the logical tiles i', i' + G', ... of ITS XCD's contiguous
// range (workgroup i runs on XCD i % 8: the tiles the CUs of an XCD work on at a time are neighbours -- a few tile rows of one image --
// so their halos come out of that XCD's L2 instead of being fetched once per XCD), and the first chunk of the NEXT tile is requested
// before the 81 result planes of the current one are stored.
__global__ __launch_bounds__(NT4, 4) void corr81_fwd4_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                          float* __restrict__ out, float* __restrict__ part, int B, int C,
                                                          int H, int W, int nslice, int cps, int otot, int ooff, float slope,
                                                          int ntx, int nty, int ntiles) {
    __shared__ __attribute__((aligned(16))) float s1[CC * TH4 * TW4];
    __shared__ __attribute__((aligned(16))) float s2[CC * F2H4 * F2W4];
    const int tid = threadIdx.x;
    // slot = 8 lanes = one tile row of ONE vertical displacement (pixels 4 tx ... 4 tx + 3 per lane): 9 x 7 = 63 of the 64 slots work,
    // the last one repeats slot 62 and stores nothing
    const int slot = tid >> 3, tx = tid & 7, uu = min(slot, 9 * TH4 - 1), dyi = uu / TH4, ty = uu - dyi * TH4;
    const bool live = slot < 9 * TH4;
    const size_t plane = (size_t)H * W;
    // XCD x owns the logical tiles [x Q, (x + 1) Q); its workgroups (blockIdx.x >> 3 = 0 .. G8 - 1) take every G8-th of them
    const int Q = (ntiles + 7) / 8, G8 = (gridDim.x + 7) / 8, xcd = blockIdx.x & 7;
    const int t_end = min(ntiles, (xcd + 1) * Q);
    struct Tile { int x0, y0, c_begin, c_end, b, slice; };
    auto tile_of = [&](int t) {
        Tile T;
        const int tx_ = t % ntx, r = t / ntx;
        const int ty_ = r % nty, z = r / nty;                              // z = b * nslice + slice
        T.b = z / nslice; T.slice = z - T.b * nslice;
        T.x0 = tx_ * TW4; T.y0 = ty_ * TH4;
        T.c_begin = T.slice * cps; T.c_end = min(C, T.c_begin + cps);
        return T;
    };

    // staging: float4 item i = tid + q NT -> (channel in chunk, halo row, column group); decoded on the fly (constant divisors) instead
    // of held in 21 registers.  A group of four pixels is inside the image or outside it as a whole (W % 4 == 0, tile origin a
    // multiple of 32).
    v4f r2[Q2], r1[Q1];                          // (native vectors: arrays of HIP's float4 struct are copied by memcpy and end up in scratch)
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](const Tile& T, int cb, int nc) {
        // (uniform 64-bit base of the chunk + a 32-bit offset per lane: seven 64-bit addresses held across the multiply phase were what
        //  pushed the kernel past the 128 registers that let two workgroups share a CU)
        const float* f1b = f1 + ((size_t)T.b * C + cb) * plane;
        const float* f2b = f2 + ((size_t)T.b * C + cb) * plane;
        int tv = tid;
        asm volatile("" : "+v"(tv));                 // (opaque: the item decode below is recomputed per fetch, ~100 VALU instructions, instead of being
                                                     //  hoisted out of the tile loop into ~30 registers that then spill)
#pragma unroll
        for (int q = 0; q < Q2; ++q) {
            const int i = tv + q * NT4;
            const int c = i / (F2H4 * (F2W4 / 4)), rem = i - c * (F2H4 * (F2W4 / 4));
            const int ly = rem / (F2W4 / 4), lx = 4 * (rem - ly * (F2W4 / 4));
            const int gy = T.y0 - 4 + ly, gx = T.x0 - 4 + lx;
#ifdef ISLAM_CORR_STAMPS
            const bool ok = !(islam_corr_dbg & 2) && c < nc && gy >= 0 && gy < H && gx >= 0 && gx < W;
#else
            const bool ok = c < nc && gy >= 0 && gy < H && gx >= 0 && gx < W;          // (c < nc <= CC also bounds i)
#endif
            r2[q] = ok ? *reinterpret_cast<const v4f*>(f2b + (unsigned)(c * (int)plane + gy * W + gx)) : zero4;
        }
#pragma unroll
        for (int q = 0; q < Q1; ++q) {
            const int i = tv + q * NT4;
            const int c = i / (TH4 * (TW4 / 4)), rem = i - c * (TH4 * (TW4 / 4));
            const int ly = rem / (TW4 / 4), lx = 4 * (rem - ly * (TW4 / 4));
            const int gy = T.y0 + ly, gx = T.x0 + lx;
#ifdef ISLAM_CORR_STAMPS
            const bool ok = !(islam_corr_dbg & 2) && c < nc && gy < H && gx < W;
#else
            const bool ok = c < nc && gy < H && gx < W;
#endif
            r1[q] = ok ? *reinterpret_cast<const v4f*>(f1b + (unsigned)(c * (int)plane + gy * W + gx)) : zero4;
        }
    };
    auto commit = [&]() {                        // item i lives at float4 slot i of its tile: both tiles are dense [c][row][col] arrays
#pragma unroll
        for (int q = 0; q < Q2; ++q)
            if (tid + q * NT4 < CC * F2H4 * (F2W4 / 4)) *reinterpret_cast<v4f*>(&s2[4 * (tid + q * NT4)]) = r2[q];
#pragma unroll
        for (int q = 0; q < Q1; ++q)
            if (tid + q * NT4 < CC * TH4 * (TW4 / 4)) *reinterpret_cast<v4f*>(&s1[4 * (tid + q * NT4)]) = r1[q];
    };

    // [measured on one box, level-2 call, as independent workgroups: 72.5 us for the one-pixel-pair kernel, 39.1 us for this lane
    //  layout; capped at 96 VGPRs for two workgroups per CU (no register prefetch, 60 bytes of scratch) it is SLOWER (49.6 vs 41.5 us
    //  on another box); the 36 FMAs as 16 v_pk_fma_f32 + 4 v_fma_f32: 39.1 us, no change -- the kernel is not VALU-bound]
    int t = xcd * Q + (blockIdx.x >> 3);
    if (t >= t_end) return;
#ifdef ISLAM_CORR_STAMPS
    const int dbg = islam_corr_dbg;
    const bool stamp = blockIdx.x == 0 && tid == 0;
    long long st_ev[62];
    int st_n = 0;
    const long long st_t0 = wall_clock64();
#endif
    Tile T = tile_of(t);
    fetch(T, T.c_begin, min(CC, T.c_end - T.c_begin));
    CSTAMP();                                        // 0: first fetch issued
    float acc[4][9];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[p][i] = 0.f;
    // the 9 x 16-byte stores of a finished tile; they zero the accumulators behind them
    auto store = [&](const Tile& S) {
        const int gy = S.y0 + ty, gx = S.x0 + 4 * tx;
#ifdef ISLAM_CORR_STAMPS
        if (live && gy < H && gx < W && (!(dbg & 1) || acc[0][0] == 12345.678f)) {
#else
        if (live && gy < H && gx < W) {
#endif
            const float sc = nslice == 1 ? 1.0f / (float)C : 1.0f;          // the mean over channels (correlation.py:97-99)
            float* ob = (nslice == 1 ? out + ((size_t)S.b * otot + ooff + (size_t)dyi * 9) * plane
                                     : part + (size_t)S.slice * B * 81 * plane + ((size_t)S.b * 81 + (size_t)dyi * 9) * plane) + (size_t)gy * W + gx;
            const float sl = nslice == 1 ? slope : 1.0f;
            auto act = [&](float v) { return v > 0.0f ? v : v * sl; };
#pragma unroll
            for (int dx = 0; dx < 9; ++dx)
                *reinterpret_cast<v4f*>(ob + (size_t)dx * plane) = v4f{act(acc[0][dx] * sc), act(acc[1][dx] * sc), act(acc[2][dx] * sc), act(acc[3][dx] * sc)};
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 9; ++i) acc[p][i] = 0.f;
    };
    for (;;) {
        const int tn = t + G8;
        const bool more = tn < t_end;
        Tile Tn = T;
        if (more) Tn = tile_of(tn);
        for (int cb = T.c_begin; cb < T.c_end; cb += CC) {
            const int nc = min(CC, T.c_end - cb);
            __syncthreads();                         // previous chunk fully consumed
            CSTAMP();                                // per chunk: barrier 1 passed
            commit();
            CSTAMP();                                //            loads arrived, LDS writes issued
            __syncthreads();
            CSTAMP();                                //            barrier 2 passed
            if (cb + CC < T.c_end) fetch(T, cb + CC, min(CC, T.c_end - cb - CC));      // in flight while this chunk is consumed
            else if (more) fetch(Tn, Tn.c_begin, min(CC, Tn.c_end - Tn.c_begin));      // ... and across the tile boundary
            CSTAMP();                                //            next fetch issued
            const float* p1 = &s1[ty * TW4 + 4 * tx];
            const float* p2 = &s2[(ty + dyi) * F2W4 + 4 * tx];
            struct Ops { v4f a, v0, v1, v2; };
            auto ld = [&](int c) {
                Ops o;
                o.a = *reinterpret_cast<const v4f*>(p1 + c * (TH4 * TW4));
                o.v0 = *reinterpret_cast<const v4f*>(p2 + c * (F2H4 * F2W4));
                o.v1 = *reinterpret_cast<const v4f*>(p2 + c * (F2H4 * F2W4) + 4);
                o.v2 = *reinterpret_cast<const v4f*>(p2 + c * (F2H4 * F2W4) + 8);
                return o;
            };
            auto mac = [&](const Ops& o) {
                const float r[12] = {o.v0.x, o.v0.y, o.v0.z, o.v0.w, o.v1.x, o.v1.y, o.v1.z, o.v1.w, o.v2.x, o.v2.y, o.v2.z, o.v2.w};
                const float av[4] = {o.a.x, o.a.y, o.a.z, o.a.w};
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int dx = 0; dx < 9; ++dx) acc[p][dx] = fmaf(av[p], r[p + dx], acc[p][dx]);
            };
            for (int c = 0; c < nc; ++c) mac(ld(c));
            CSTAMP();                                //            multiplied
        }
        store(T);
        CSTAMP();                                    // per tile: stores issued
        if (!more) break;
        t = tn;
        T = Tn;
    }
#ifdef ISLAM_CORR_STAMPS
    if (stamp) {
        islam_corr_stamps_buf[0] = st_n;
        for (int q = 0; q < st_n; ++q) islam_corr_stamps_buf[1 + q] = st_ev[q];
        islam_corr_stamps_buf[63] = wall_clock64() - st_t0;
    }
#endif
}

__global__ __launch_bounds__(256) void corr81_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n,
                                                             int nslice, float inv_c, size_t img /* 81 * H * W */, int otot, int ooff,
                                                             float slope) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = part[i];
    for (int k = 1; k < nslice; ++k) s += part[(size_t)k * n + i];       // fixed order: deterministic
    s *= inv_c;
    const size_t b = i / img, rem = i - b * img;
    out[(b * otot + ooff) * (img / 81) + rem] = s > 0.0f ? s : s * slope;
}

// gradient w.r.t. the first input:  g1[b,c,y,x] = (1/C) sum_{p,o} gout[b,(p+4)*9+(o+4),y,x] * f2[b,c,y+p,x+o]
// gradient w.r.t. the second input: g2[b,c,y,x] = (1/C) sum_{p,o} gout[b,op,y-p,x-o]    * f1[b,c,y-p,x-o]
// A workgroup owns a 32x8 pixel tile and 8 channels; the 81 gout planes of the tile (with halo for g2)
// are staged through LDS nine at a time (one dy row of displacements per pass).
constexpr int BW = 32, BH = 8, BC = 8;

template <bool SECOND>
__global__ __launch_bounds__(256) void corr81_bwd_kernel(const float* __restrict__ fin, const float* __restrict__ gout,
                                                          float* __restrict__ gin, int C, int H, int W) {
    // FIRST : needs gout at the pixel itself (no halo) and f2 with halo
    // SECOND: needs gout*f1 at (y-p, x-o): stage the product plane with halo
    __shared__ float sg[9][BH + 8][BW + 8];      // one dy row of gout planes (SECOND: with halo)
    __shared__ float sf[BC][BH + 8][BW + 8];     // input feature tile with halo
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int b = blockIdx.z / ((C + BC - 1) / BC);
    const int cb = (blockIdx.z % ((C + BC - 1) / BC)) * BC;
    const int x0 = blockIdx.x * BW, y0 = blockIdx.y * BH;
    const size_t plane = (size_t)H * W;
    const int nc = min(BC, C - cb);
    for (int i = threadIdx.x; i < nc * (BH + 8) * (BW + 8); i += 256) {
        const int c = i / ((BH + 8) * (BW + 8)), rem = i - c * ((BH + 8) * (BW + 8));
        const int ly = rem / (BW + 8), lx = rem - ly * (BW + 8);
        const int gy = y0 - 4 + ly, gx = x0 - 4 + lx;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = fin[((size_t)b * C + cb + c) * plane + (size_t)gy * W + gx];
        sf[c][ly][lx] = v;
    }
    float acc[BC];
#pragma unroll
    for (int c = 0; c < BC; ++c) acc[c] = 0.f;
    for (int p = 0; p < 9; ++p) {
        __syncthreads();
        for (int i = threadIdx.x; i < 9 * (BH + 8) * (BW + 8); i += 256) {
            const int o = i / ((BH + 8) * (BW + 8)), rem = i - o * ((BH + 8) * (BW + 8));
            const int ly = rem / (BW + 8), lx = rem - ly * (BW + 8);
            const int gy = y0 - 4 + ly, gx = x0 - 4 + lx;
            float v = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = gout[((size_t)b * 81 + p * 9 + o) * plane + (size_t)gy * W + gx];
            sg[o][ly][lx] = v;
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 9; ++o) {
            if (!SECOND) {
                const float g = sg[o][ty + 4][tx + 4];
#pragma unroll
                for (int c = 0; c < BC; ++c) acc[c] = fmaf(g, sf[c][ty + p][tx + o], acc[c]);
            } else {
                // source pixel (y - (p-4), x - (o-4)) -> local (ty + 4 - (p-4), tx + 4 - (o-4)) = (ty + 8 - p, tx + 8 - o)
                const float g = sg[o][ty + 8 - p][tx + 8 - o];
#pragma unroll
                for (int c = 0; c < BC; ++c) acc[c] = fmaf(g, sf[c][ty + 8 - p][tx + 8 - o], acc[c]);
            }
        }
    }
    const int gy = y0 + ty, gx = x0 + tx;
    if (gy < H && gx < W) {
        const float fc = (float)C;
        for (int c = 0; c < nc; ++c) gin[((size_t)b * C + cb + c) * plane + (size_t)gy * W + gx] = acc[c] / fc;
    }
}

// PWCDCNet.warp.  The coordinate pipeline restates torch's grid_sample(align_corners=True) in the
// same float32 operation order (normalise to [-1,1], un-normalise, floor, corner weights
// nw=(x1-ix)(y1-iy) ...), so fused multiply-add contraction is disabled here.
constexpr int WCH = 16;       // channels per lane

__global__ __launch_bounds__(256) void warp_mask_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                         float scale, float* __restrict__ out, int C, int H, int W) {
#pragma clang fp contract(off)
    const int ngrp = (C + WCH - 1) / WCH;
    const int b = blockIdx.z / ngrp, c0 = (blockIdx.z % ngrp) * WCH;
    // pixels in row-major order, 256 per workgroup: with 64 x 4-pixel blocks a 160-pixel row left a sixth of the lanes idle and the
    // level-2 call (B = 8) had 1344 workgroups for the 1280 a chip holds at once at this kernel's ~100 registers -- a second round for
    // 5 % of the work; 1120 workgroups fit one
    const int pidx = blockIdx.x * 256 + threadIdx.x;
    if (pidx >= H * W) return;
    const int py = pidx / W, px = pidx - py * W;
    const size_t plane = (size_t)H * W;
    const size_t pix = (size_t)pidx;
    const float fx = flow[((size_t)b * 2 + 0) * plane + pix] * scale;
    const float fy = flow[((size_t)b * 2 + 1) * plane + pix] * scale;
    const float vx = (float)px + fx, vy = (float)py + fy;
    const float gx = 2.0f * vx / (float)(W > 1 ? W - 1 : 1) - 1.0f;
    const float gy = 2.0f * vy / (float)(H > 1 ? H - 1 : 1) - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    const float nw = (x1f - ix) * (y1f - iy), ne = (ix - x0f) * (y1f - iy);
    const float sw = (x1f - ix) * (iy - y0f), se = (ix - x0f) * (iy - y0f);
    // NaN/inf coordinates fail every comparison -> no valid corner -> output 0 like grid_sample's bounds test
    const bool vx0 = x0f >= 0.f && x0f <= (float)(W - 1), vx1 = x1f >= 0.f && x1f <= (float)(W - 1);
    const bool vy0 = y0f >= 0.f && y0f <= (float)(H - 1), vy1 = y1f >= 0.f && y1f <= (float)(H - 1);
    const bool v00 = vx0 && vy0, v01 = vx1 && vy0, v10 = vx0 && vy1, v11 = vx1 && vy1;
    const int xi0 = vx0 ? (int)x0f : 0, xi1 = vx1 ? (int)x1f : 0, yi0 = vy0 ? (int)y0f : 0, yi1 = vy1 ? (int)y1f : 0;
    float m = 0.f;
    if (v00) m += nw;
    if (v01) m += ne;
    if (v10) m += sw;
    if (v11) m += se;
    const float mask = (m < 0.9999f) ? 0.f : 1.f;      // PWCNet.py:203-204
    const size_t o00 = (size_t)yi0 * W + xi0, o01 = (size_t)yi0 * W + xi1, o10 = (size_t)yi1 * W + xi0, o11 = (size_t)yi1 * W + xi1;
    const float* xb = x + ((size_t)b * C + c0) * plane;
    float* ob = out + ((size_t)b * C + c0) * plane + pix;
    const int nc = min(WCH, C - c0);
    // all taps of the channel group are requested before the first is used (loads are unconditional on valid
    // addresses; invalid corners are dropped by the selects)
    float t00[WCH], t01[WCH], t10[WCH], t11[WCH];
#pragma unroll
    for (int c = 0; c < WCH; ++c) {
        const float* p = xb + (size_t)(c < nc ? c : 0) * plane;
        t00[c] = p[o00]; t01[c] = p[o01]; t10[c] = p[o10]; t11[c] = p[o11];
    }
#pragma unroll
    for (int c = 0; c < WCH; ++c) {
        if (c < nc) {
            float s = 0.f;
            if (v00) s += t00[c] * nw;
            if (v01) s += t01[c] * ne;
            if (v10) s += t10[c] * sw;
            if (v11) s += t11[c] * se;
            ob[(size_t)c * plane] = s * mask;
        }
    }
}

// Backward of warp_mask_kernel (what autograd derives for PWCNet.py:195-206): the mask is piecewise constant, so
// d out = mask * d grid_sample.  gx (same shape as x) and gflow (B,2,H,W) must be zero-initialised: the four taps scatter
// with atomics (like torch's own grid_sampler backward) and the channel groups accumulate into gflow.
__global__ __launch_bounds__(256) void warp_mask_bwd_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                             float scale, const float* __restrict__ gout, float* gx_,
                                                             float* gflow, int C, int H, int W) {
    const int ngrp = (C + WCH - 1) / WCH;
    const int b = blockIdx.z / ngrp, c0 = (blockIdx.z % ngrp) * WCH;
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= W || py >= H) return;
    const size_t plane = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;
    float ix, iy;
    {
#pragma clang fp contract(off)
        const float fx = flow[((size_t)b * 2 + 0) * plane + pix] * scale;
        const float fy = flow[((size_t)b * 2 + 1) * plane + pix] * scale;
        const float vx = (float)px + fx, vy = (float)py + fy;
        const float gxn = 2.0f * vx / (float)(W > 1 ? W - 1 : 1) - 1.0f;
        const float gyn = 2.0f * vy / (float)(H > 1 ? H - 1 : 1) - 1.0f;
        ix = ((gxn + 1.0f) / 2.0f) * (float)(W - 1);
        iy = ((gyn + 1.0f) / 2.0f) * (float)(H - 1);
    }
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    const float nw = (x1f - ix) * (y1f - iy), ne = (ix - x0f) * (y1f - iy);
    const float sw = (x1f - ix) * (iy - y0f), se = (ix - x0f) * (iy - y0f);
    const bool vx0 = x0f >= 0.f && x0f <= (float)(W - 1), vx1 = x1f >= 0.f && x1f <= (float)(W - 1);
    const bool vy0 = y0f >= 0.f && y0f <= (float)(H - 1), vy1 = y1f >= 0.f && y1f <= (float)(H - 1);
    const bool v00 = vx0 && vy0, v01 = vx1 && vy0, v10 = vx0 && vy1, v11 = vx1 && vy1;
    const int xi0 = vx0 ? (int)x0f : 0, xi1 = vx1 ? (int)x1f : 0, yi0 = vy0 ? (int)y0f : 0, yi1 = vy1 ? (int)y1f : 0;
    float m = 0.f;
    if (v00) m += nw;
    if (v01) m += ne;
    if (v10) m += sw;
    if (v11) m += se;
    if (m < 0.9999f) return;                         // masked pixel: no gradient
    const size_t o00 = (size_t)yi0 * W + xi0, o01 = (size_t)yi0 * W + xi1, o10 = (size_t)yi1 * W + xi0, o11 = (size_t)yi1 * W + xi1;
    const int nc = min(WCH, C - c0);
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < nc; ++c) {
        const size_t cp = ((size_t)b * C + c0 + c) * plane;
        const float g = gout[cp + pix];
        const float* p = x + cp;
        float* q = gx_ + cp;
        if (v00) { atomicAdd(q + o00, g * nw); const float t = p[o00]; gix -= t * (y1f - iy) * g; giy -= t * (x1f - ix) * g; }
        if (v01) { atomicAdd(q + o01, g * ne); const float t = p[o01]; gix += t * (y1f - iy) * g; giy -= t * (ix - x0f) * g; }
        if (v10) { atomicAdd(q + o10, g * sw); const float t = p[o10]; gix -= t * (iy - y0f) * g; giy += t * (x1f - ix) * g; }
        if (v11) { atomicAdd(q + o11, g * se); const float t = p[o11]; gix += t * (iy - y0f) * g; giy += t * (ix - x0f) * g; }
    }
    // d ix / d flow_x = scale * (W-1)/max(W-1,1), same for y
    const float kx = W > 1 ? scale : 0.f, ky = H > 1 ? scale : 0.f;
    atomicAdd(gflow + ((size_t)b * 2 + 0) * plane + pix, gix * kx);
    atomicAdd(gflow + ((size_t)b * 2 + 1) * plane + pix, giy * ky);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// ConvTranspose2d(C -> 2, kernel 4, stride 2, padding 1) + bias: PWC-Net's `deconv%d` (2 -> 2 channels: the up-sampled flow) and
// `upfeat%d` (the whole DenseNet concatenation, up to 565 channels, -> 2) (Network/PWC/PWCNet.py:61-63, 260-291).  With two output
// channels this is a memory-bound reduction over the input channels, not a GEMM: MIOpen ran the large ones as Winograd / implicit-GEMM
// backward-data kernels at 0.4 TB/s (184 + 126 us at the 56x80 level, B = 8).  fp32 NCHW in and out, exact fp32 FMAs.
// Output pixel (2y + a, 2x + c) = b[o] + sum_i sum_{r,s in {0,1}} x[i, y-1+a+r, x-1+c+s] W[i, o, 3-2r-a, 3-2s-c]: a thread owns input
// pixel (y, x), i.e. the 2x2 output block x 2 channels (8 sums) over its 3x3 input neighbourhood; the sixteen waves of a workgroup split the
// channels (i = wave, wave + 16, ...: a channel's 32 weights are wave-uniform scalar loads) and add their sums in wave order (deterministic).
constexpr int UP2_WAVES = 16;                        // waves of a workgroup = channel classes (i mod 16): the reduction is latency-bound per wave
//
// HEAD: the same pass also evaluates PWC-Net's flow head `predict_flow%d` = Conv2d(C -> 2, kernel 3, padding 1) + bias (PWCNet.py:112 ff.,
// used right before `upfeat%d` on the SAME tensor): flow[o, y, x] = b[o] + sum_i sum_{dy,dx} x[i, y-1+dy, x-1+dx] Wf[o, i, dy, dx] is another
// two sums over the nine values the lane already holds -- 18 more FMAs per channel instead of a second pass over up to 565 channels
// (the head ran as a 64-output-channel matrix-core tile with 2 live channels: 54-180 us per level).  wf: [C][2][9] (re-packed on the host).
// UP = false: the head alone (level 2 has no up-sampling behind it).
template <bool HEAD, bool UP>
__global__ __launch_bounds__(64 * UP2_WAVES) void deconv4x4s2_to2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                         const float* __restrict__ bias, float* __restrict__ y, int C, int H, int W,
                                                                         int ytot, int coff, const float* __restrict__ wf,
                                                                         const float* __restrict__ bf, float* __restrict__ flow) {
    constexpr int NS = (UP ? 8 : 0) + (HEAD ? 2 : 0);
    __shared__ float red[UP2_WAVES - 1][64][NS + 1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: weights by scalar loads)
    const int px = blockIdx.x * 64 + lane, py = blockIdx.y, b = blockIdx.z;
    const bool on = px < W;
    float acc[NS];
#pragma unroll
    for (int e = 0; e < NS; ++e) acc[e] = 0.0f;
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * C * plane;
    bool vy[3], vx[3];
    int oy[3], ox[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int yy = py - 1 + d, xx = px - 1 + d;
        vy[d] = yy >= 0 && yy < H; vx[d] = on && xx >= 0 && xx < W;
        oy[d] = min(max(yy, 0), H - 1) * W; ox[d] = min(max(xx, 0), W - 1);
    }
    auto load9 = [&](int i, float (&v)[3][3]) {              // (clamped addresses: the same nine loads on every lane)
        const float* xc = xb + (size_t)min(i, C - 1) * plane;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) v[dy][dx] = xc[oy[dy] + ox[dx]];
    };
    auto fma_all = [&](int i, const float (&v)[3][3]) {
        if (i >= C) return;                                   // (wave-uniform)
        float m[3][3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) m[dy][dx] = (vy[dy] && vx[dx]) ? v[dy][dx] : 0.0f;
        if constexpr (UP) {
            const float* wc = w + (size_t)i * 32;             // [o][ky][kx], wave-uniform: scalar loads
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int r = 0; r < 2; ++r)
#pragma unroll
                            for (int sx = 0; sx < 2; ++sx)
                                acc[o * 4 + a * 2 + c] = fmaf(m[a + r][c + sx], wc[o * 16 + (3 - 2 * r - a) * 4 + (3 - 2 * sx - c)], acc[o * 4 + a * 2 + c]);
        }
        if constexpr (HEAD) {
            const float* wh = wf + (size_t)i * 18;            // [o][dy][dx]
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) acc[(UP ? 8 : 0) + o] = fmaf(m[dy][dx], wh[o * 9 + dy * 3 + dx], acc[(UP ? 8 : 0) + o]);
        }
    };
    // two channels per iteration, the next pair's eighteen loads in flight while this pair is accumulated
    float v0[3][3], v1[3][3], n0[3][3], n1[3][3];
    load9(wave, v0);
    load9(wave + UP2_WAVES, v1);
    for (int i = wave; i < C; i += 2 * UP2_WAVES) {
        load9(i + 2 * UP2_WAVES, n0);
        load9(i + 3 * UP2_WAVES, n1);
        fma_all(i, v0);
        fma_all(i + UP2_WAVES, v1);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) { v0[dy][dx] = n0[dy][dx]; v1[dy][dx] = n1[dy][dx]; }
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < NS; ++e) red[wave - 1][lane][e] = acc[e];
    }
    __syncthreads();
    if (wave == 0 && on) {
#pragma unroll
        for (int e = 0; e < NS; ++e)
            for (int q = 0; q < UP2_WAVES - 1; ++q) acc[e] += red[q][lane][e];
        if constexpr (UP) {
            const int Ho = 2 * H, Wo = 2 * W;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                float* yo = y + ((size_t)b * ytot + coff + o) * Ho * Wo;
                const float bo = bias ? bias[o] : 0.0f;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float2 out;
                    out.x = acc[o * 4 + a * 2] + bo; out.y = acc[o * 4 + a * 2 + 1] + bo;
                    *reinterpret_cast<float2*>(yo + (size_t)(2 * py + a) * Wo + 2 * px) = out;
                }
            }
        }
        if constexpr (HEAD) {
#pragma unroll
            for (int o = 0; o < 2; ++o) flow[((size_t)b * 2 + o) * plane + (size_t)py * W + px] = acc[(UP ? 8 : 0) + o] + (bf ? bf[o] : 0.0f);
        }
    }
}

// The same sums with FOUR pixels per lane (W % 4 == 0): a lane owns pixels (y, x0 .. x0+3); per channel and input row it requests one
// 16-byte vector (columns x0 .. x0+3) + the two neighbours x0-1, x0+4 -- 9 requests per 4 pixels instead of 36.  With one pixel per
// lane the kernel was bound by the request rate of the vector memory pipe (nine 4-byte requests per pixel and channel: 0.8-1.3 TB/s).
// Workgroup = 8 waves (channel classes i mod 8) x a 64 x 4 pixel block (16 lanes along x, 4 along y).
constexpr int HU_WAVES = 8;
template <bool HEAD, bool UP>
__global__ __launch_bounds__(64 * HU_WAVES) void head_up4_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                 float* __restrict__ y, int C, int H, int W, int ytot, int coff,
                                                                 const float* __restrict__ wf, const float* __restrict__ bf, float* __restrict__ flow) {
    constexpr int NS = (UP ? 8 : 0) + (HEAD ? 2 : 0), HO = UP ? 8 : 0;
    __shared__ float red[HU_WAVES - 1][64][NS + 1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = blockIdx.x * 64 + (lane & 15) * 4, py = blockIdx.y * 4 + (lane >> 4), b = blockIdx.z;
    const bool on = x0 < W && py < H;                        // (W % 4 == 0: the four pixels are inside together)
    float acc[4][NS];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < NS; ++e) acc[j][e] = 0.0f;
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * C * plane;
    bool vy[3];
    int oy[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int yy = py - 1 + d;
        vy[d] = on && yy >= 0 && yy < H;
        oy[d] = min(max(yy, 0), H - 1) * W;
    }
    const int xq = min(x0, W - 4), xl = max(xq - 1, 0), xr = min(x0 + 4, W - 1);
    const bool vl = x0 >= 1, vr = x0 + 4 < W;
    auto load18 = [&](int i, float (&v)[3][6]) {             // (clamped addresses: the same nine requests on every lane)
        const float* xc = xb + (size_t)min(i, C - 1) * plane;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float4 q = *reinterpret_cast<const float4*>(xc + oy[d] + xq);
            v[d][0] = xc[oy[d] + xl]; v[d][1] = q.x; v[d][2] = q.y; v[d][3] = q.z; v[d][4] = q.w; v[d][5] = xc[oy[d] + xr];
        }
    };
    auto fma_all = [&](int i, const float (&v)[3][6]) {
        if (i >= C) return;                                   // (wave-uniform)
        float m[3][6];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int k = 0; k < 6; ++k) m[d][k] = (vy[d] && (k == 0 ? vl : k == 5 ? vr : true)) ? v[d][k] : 0.0f;
        if constexpr (UP) {
            const float* wc = w + (size_t)i * 32;             // [o][ky][kx], wave-uniform: scalar loads
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int r = 0; r < 2; ++r)
#pragma unroll
                            for (int sx = 0; sx < 2; ++sx) {
                                const float wv = wc[o * 16 + (3 - 2 * r - a) * 4 + (3 - 2 * sx - c)];
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc[j][o * 4 + a * 2 + c] = fmaf(m[a + r][j + c + sx], wv, acc[j][o * 4 + a * 2 + c]);
                            }
        }
        if constexpr (HEAD) {
            const float* wh = wf + (size_t)i * 18;            // [o][dy][dx]
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float wv = wh[o * 9 + dy * 3 + dx];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j][HO + o] = fmaf(m[dy][j + dx], wv, acc[j][HO + o]);
                    }
        }
    };
    float v[3][6], n[3][6];
    load18(wave, v);
    for (int i = wave; i < C; i += HU_WAVES) {               // the next channel's nine requests in flight while this one is accumulated
        load18(i + HU_WAVES, n);
        fma_all(i, v);
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int k = 0; k < 6; ++k) v[d][k] = n[d][k];
    }
    for (int j = 0; j < 4; ++j) {                             // the waves' sums, added in wave order (deterministic), one pixel column at a time
        if (wave > 0) {
#pragma unroll
            for (int e = 0; e < NS; ++e) red[wave - 1][lane][e] = acc[j][e];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int e = 0; e < NS; ++e) {
                float t = acc[j][e];
                for (int q = 0; q < HU_WAVES - 1; ++q) t += red[q][lane][e];
                acc[j][e] = t;
            }
        }
        __syncthreads();
    }
    if (wave != 0 || !on) return;
    if constexpr (UP) {
        const int Ho = 2 * H, Wo = 2 * W;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float* yo = y + ((size_t)b * ytot + coff + o) * Ho * Wo;
            const float bo = bias ? bias[o] : 0.0f;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float4* dst = reinterpret_cast<float4*>(yo + (size_t)(2 * py + a) * Wo + 2 * x0);
                dst[0] = make_float4(acc[0][o * 4 + a * 2] + bo, acc[0][o * 4 + a * 2 + 1] + bo, acc[1][o * 4 + a * 2] + bo, acc[1][o * 4 + a * 2 + 1] + bo);
                dst[1] = make_float4(acc[2][o * 4 + a * 2] + bo, acc[2][o * 4 + a * 2 + 1] + bo, acc[3][o * 4 + a * 2] + bo, acc[3][o * 4 + a * 2 + 1] + bo);
            }
        }
    }
    if constexpr (HEAD) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const float bo = bf ? bf[o] : 0.0f;
            *reinterpret_cast<float4*>(flow + ((size_t)b * 2 + o) * plane + (size_t)py * W + x0) =
                make_float4(acc[0][HO + o] + bo, acc[1][HO + o] + bo, acc[2][HO + o] + bo, acc[3][HO + o] + bo);
        }
    }
}

template <bool HEAD, bool UP>
static void launch_head_up(const float* x, const float* w, const float* bias, float* y, int C, int H, int W, int ytot, int coff, const float* wf,
                           const float* bf, float* flow, int B, hipStream_t s) {
    // four pixels per lane only on large maps (level 2: 112x160): below, a level has too few pixel blocks to hide the per-channel round trip
    // and the one-pixel kernel's 4x more waves win (upfeat4, 28x40: 31 vs 134 us; upfeat3, 56x80: 111 vs 136 us; head of level 2: 254 vs 186 us)
    const bool quad = (size_t)H * W >= 8192 && (W & 3) == 0 && ((uintptr_t)x & 15) == 0 && (!UP || (((uintptr_t)y & 15) == 0)) && (!HEAD || (((uintptr_t)flow & 15) == 0));
    if (quad)
        hipLaunchKernelGGL((head_up4_kernel<HEAD, UP>), dim3((W + 63) / 64, (H + 3) / 4, B), dim3(64 * HU_WAVES), 0, s, x, w, bias, y, C, H, W, ytot, coff,
                           wf, bf, flow);
    else
        hipLaunchKernelGGL((deconv4x4s2_to2_kernel<HEAD, UP>), dim3((W + 63) / 64, H, B), dim3(64 * UP2_WAVES), 0, s, x, w, bias, y, C, H, W, ytot, coff,
                           wf, bf, flow);
}

#ifdef ISLAM_CORR_STAMPS
extern "C" int islam_corr_dbg_set(int v) { return hipMemcpyToSymbol(HIP_SYMBOL(islam_corr_dbg), &v, sizeof(int)) == hipSuccess ? 0 : 1; }
extern "C" int islam_corr_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(islam_corr_stamps_buf), sizeof(long long) * 64) == hipSuccess ? 0 : 1;
}
#endif

extern "C" {

// which kernel, how many channel slices (small pyramid levels: too few tiles for 256 CUs), channels per slice -- ONE place, so that
// the scratch a caller sizes with islam_corr81_scratch_bytes is the scratch the launch uses
static void corr81_plan(int B, int C, int H, int W, bool* fwd4, int* nslice, int* cps) {
    static const bool corr4 = [] { const char* e = std::getenv("ISLAM_CORR4"); return !(e && e[0] == '0'); }();   // 0: the one-pixel-pair kernel (A/B runs)
    *fwd4 = corr4 && (W & 3) == 0 && H * W >= 1024;                      // small maps keep the 32 x 4 tile (more workgroups)
    const int th = *fwd4 ? TH4 : TH;
    const int tiles = ((W + TW - 1) / TW) * ((H + th - 1) / th) * B;
    const int chunks = (C + CC - 1) / CC;
    // workgroups wanted before the channels are split into slices (partial sums + a reduce pass: 4 x the output traffic of the level-3
    // call).  Past ~256 workgroups of the four-pixel kernel slicing only adds traffic -- B = 8, level 3 (C 64, 56 x 80) 39.3 -> 22.5 us,
    // level 4 (C 96, 28 x 40) 29.1 -> 21.4 us with 256 instead of 1024 (round 4, one workgroup per CU); with the 7-row tiles and two
    // workgroups per CU of round 6: 19.1 / 18.7 us at 128 or 256, 30.8 / 23.0 us at 512; the small-map kernel keeps 1024 (16.3 -> 18.6 us
    // at 128).  ISLAM_CORR_SLICE_TARGET overrides both (A/B runs).
    static const int forced = [] { const char* e = std::getenv("ISLAM_CORR_SLICE_TARGET"); return e && std::atoi(e) > 0 ? std::atoi(e) : 0; }();
    const int target = forced ? forced : (*fwd4 ? 256 : 1024);
    int ns = std::min(chunks, std::max(1, target / std::max(tiles, 1)));
    *cps = ((chunks + ns - 1) / ns) * CC;                                // channels per slice (whole chunks)
    *nslice = (C + *cps - 1) / *cps;
}

size_t islam_corr81_scratch_bytes(int B, int C, int H, int W) {
    bool fwd4;
    int nslice, cps;
    corr81_plan(B, C, H, W, &fwd4, &nslice, &cps);
    if (nslice <= 1) return 0;
    return (size_t)nslice * B * 81 * H * W * sizeof(float);
}

static int corr81_launch(const float* f1, const float* f2, float* out, int otot, int ooff, float slope, int B, int C, int H, int W, void* scratch,
                         void* stream) {
    bool fwd4;
    int nslice, cps;
    corr81_plan(B, C, H, W, &fwd4, &nslice, &cps);
    if (nslice > 1 && scratch == nullptr) { nslice = 1; cps = ((C + CC - 1) / CC) * CC; }      // no scratch: single pass
    const bool al = ((reinterpret_cast<uintptr_t>(f1) | reinterpret_cast<uintptr_t>(f2) | reinterpret_cast<uintptr_t>(out) |
                      reinterpret_cast<uintptr_t>(scratch)) & 15) == 0;
    if (fwd4 && al) {
        // one workgroup per CU (a multiple of 8, so that every XCD gets the same number), each walking its share of the tiles
        const int ntx = (W + TW4 - 1) / TW4, nty = (H + TH4 - 1) / TH4, ntiles = ntx * nty * B * nslice;
        static const int cus = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
            n *= 2;                                                      // two workgroups of 8 waves per CU
            const char* e = std::getenv("ISLAM_CORR4_WGS");              // (A/B runs: workgroups of the launch)
            if (e && std::atoi(e) > 0) n = std::atoi(e);
            return n / 8 * 8;
        }();
        const int grid4 = std::max(8, std::min(cus, (ntiles + 7) / 8 * 8));
        hipLaunchKernelGGL(corr81_fwd4_kernel, dim3(grid4), dim3(NT4), 0, as_stream(stream), f1, f2, out, (float*)scratch, B, C, H, W, nslice, cps,
                           otot, ooff, slope, ntx, nty, ntiles);
    } else {
        dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, B * nslice), block(16, 4, 9);
        hipLaunchKernelGGL(corr81_fwd_kernel, grid, block, 0, as_stream(stream), f1, f2, out, (float*)scratch, B, C, H, W, nslice, cps, otot, ooff, slope);
    }
    if (nslice > 1) {
        const size_t n = (size_t)B * 81 * H * W;
        hipLaunchKernelGGL(corr81_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                           (const float*)scratch, out, n, nslice, 1.0f / (float)C, (size_t)81 * H * W, otot, ooff, slope);
    }
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_corr81_fwd(const float* f1, const float* f2, float* out, int B, int C, int H, int W, void* scratch, void* stream) {
    if (B < 1 || C < 1 || H < 1 || W < 1) return fail(ISLAM_EARG, "islam_corr81_fwd: bad shape (%d,%d,%d,%d)", B, C, H, W);
    return corr81_launch(f1, f2, out, 81, 0, 1.0f, B, C, H, W, scratch, stream);
}

int islam_corr81_fwd_act(const float* f1, const float* f2, float* out, int otot, int ooff, float slope, int B, int C, int H, int W, void* scratch,
                         void* stream) {
    if (B < 1 || C < 1 || H < 1 || W < 1 || ooff < 0 || ooff + 81 > otot)
        return fail(ISLAM_EARG, "islam_corr81_fwd_act: bad argument (%d,%d,%d,%d), slice %d+81 of %d", B, C, H, W, ooff, otot);
    return corr81_launch(f1, f2, out, otot, ooff, slope, B, C, H, W, scratch, stream);
}

int islam_corr81_bwd(const float* f1, const float* f2, const float* gout, float* g1, float* g2, int B, int C, int H, int W,
                     void* stream) {
    if (B < 1 || C < 1 || H < 1 || W < 1) return fail(ISLAM_EARG, "islam_corr81_bwd: bad shape (%d,%d,%d,%d)", B, C, H, W);
    dim3 grid((W + BW - 1) / BW, (H + BH - 1) / BH, B * ((C + BC - 1) / BC)), block(256);
    if (g1) hipLaunchKernelGGL(corr81_bwd_kernel<false>, grid, block, 0, as_stream(stream), f2, gout, g1, C, H, W);
    if (g2) hipLaunchKernelGGL(corr81_bwd_kernel<true>, grid, block, 0, as_stream(stream), f1, gout, g2, C, H, W);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_warp_mask(const float* x, const float* flow, float scale, float* out, int B, int C, int H, int W, void* stream) {
    if (B < 1 || C < 1 || H < 1 || W < 1) return fail(ISLAM_EARG, "islam_warp_mask: bad shape (%d,%d,%d,%d)", B, C, H, W);
    if ((long long)H * W >= (1LL << 31) - 256) return fail(ISLAM_EARG, "islam_warp_mask: image too large (%dx%d)", H, W);
    dim3 grid((unsigned)(((long long)H * W + 255) / 256), 1, B * ((C + WCH - 1) / WCH)), block(256);
    hipLaunchKernelGGL(warp_mask_kernel, grid, block, 0, as_stream(stream), x, flow, scale, out, C, H, W);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_warp_mask_bwd(const float* x, const float* flow, float scale, const float* gout, float* gx, float* gflow, int B,
                        int C, int H, int W, void* stream) {
    if (B < 1 || C < 1 || H < 1 || W < 1) return fail(ISLAM_EARG, "islam_warp_mask_bwd: bad shape (%d,%d,%d,%d)", B, C, H, W);
    dim3 grid((W + 63) / 64, (H + 3) / 4, B * ((C + WCH - 1) / WCH)), block(256);
    hipLaunchKernelGGL(warp_mask_bwd_kernel, grid, block, 0, as_stream(stream), x, flow, scale, gout, gx, gflow, C, H, W);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_deconv4x4s2_to2_f32(const float* x, const float* w, const float* bias, float* y, int ytot, int coff, int B, int C, int H, int W,
                              void* stream) {
    if (B < 1 || C < 1 || H < 1 || W < 1 || coff < 0 || coff + 2 > ytot) return fail(ISLAM_EARG, "islam_deconv4x4s2_to2_f32: bad argument (C=%d, %dx%d, slice %d+2 of %d)", C, H, W, coff, ytot);
    launch_head_up<false, true>(x, w, bias, y, C, H, W, ytot, coff, nullptr, nullptr, nullptr, B, as_stream(stream));
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_flow_head_up_f32(const float* x, const float* wf, const float* bf, float* flow, const float* wu, const float* bu, float* up, int uptot,
                           int upoff, int B, int C, int H, int W, void* stream) {
    if (B < 1 || C < 1 || H < 1 || W < 1 || !wf || !flow) return fail(ISLAM_EARG, "islam_flow_head_up_f32: bad argument (C=%d, %dx%d)", C, H, W);
    if (wu && (!up || upoff < 0 || upoff + 2 > uptot)) return fail(ISLAM_EARG, "islam_flow_head_up_f32: slice %d+2 of %d", upoff, uptot);
    if (wu) launch_head_up<true, true>(x, wu, bu, up, C, H, W, uptot, upoff, wf, bf, flow, B, as_stream(stream));
    else launch_head_up<true, false>(x, nullptr, nullptr, nullptr, C, H, W, 0, 0, wf, bf, flow, B, as_stream(stream));
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

}  // extern "C"
