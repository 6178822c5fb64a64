// The trainable pose head (reference Network/VOFlowNet.py:20-39 BasicBlock, :110-157 feature embedding of config 1, :185-194 forward_):
// forward AND backward of the whole network as two C calls that enqueue hand-written fp32 kernels -- no MIOpen / CK / ATen launch.
//
//   feat_net: conv(4,32,s2)+ReLU, 2 x conv(32,32)+ReLU, then 5 stages of BasicBlocks (64x3, 128x4, 128x6, 256x7, 256x3; the first block
//   of a stage has stride 2 and a 1x1 stride-2 convolution on its shortcut), flatten (NCHW order), two heads Linear(1536,128)+ReLU,
//   Linear(128,32)+ReLU, Linear(32,3); output cat(trans, rot).
//
// Arithmetic: exact fp32 on v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain, bit for bit; 157 TFLOP/s chip peak) -- the head is trained,
// its forward is held to 1e-5 and its gradients to 1e-4 of fp32 autograd (tests/test_pose_head_gpu.py).  Activations and gradients are
// fp32 channels-last (B,H,W,C); weights are read where torch keeps them for a channels_last module: [Cout][ky][kx][Cin].
//
// Every convolution, its data gradient and its weight gradient are ONE implicit-GEMM kernel family (conv_gemm_kernel<MODE, KC, WIDE>):
//   forward   D[pixel][co] = sum_{tap,ci}  in[pixel@tap][ci]  * w[co][tap][ci]      (+ the 1x1 stride-2 shortcut convolution as extra K)
//   dgrad     D[pixel][ci] = sum_{tap,co} (g*mask)[pixel@tap'][co] * w[co][tap][ci] (+ the shortcut's transposed 1x1 as extra K)
//   wgrad     D[co][ci]    = sum_{pixel}  (g*mask)[pixel][co] * in[pixel@tap][ci]    per tap (+ bias gradient = column sums of g*mask)
// A workgroup of four wavefronts owns a 32-channel column block: WIDE = 128 rows, one 32x32 accumulator per wave over the whole chunk;
// otherwise 32 rows, the four waves split every K chunk and their accumulators are summed through LDS in wave order.  Layers too small
// to fill 256 CUs additionally split K over workgroups (gridDim.y); partial tiles go to scratch with write-through stores, the workgroup
// that draws the tile's last ticket adds them in split order and runs the epilogue (bias, shortcut, ReLU / ReLU mask / accumulate into
// the gradient) -- a fixed summation order whichever workgroup arrives last: results are deterministic run to run (MIOpen's split-K
// kernels add with atomics; DESIGN.md section 9.6 of round 5 chased the ReLU flips that causes).
// ReLU masks are never materialised: a gradient operand is loaded together with the forward activation it belongs to and zeroed where
// that is not positive.  Global loads of chunk c+1 are in flight while chunk c is multiplied (register prefetch, two LDS buffers, one
// barrier per chunk).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/islam_hip.h"
#include "common.h"

using namespace islam;

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int TS = 256;                             // threads of the small kernels (first convolution, Linear layers)

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };

struct ConvArgs {
    // ---- the 3x3 convolution (pad 1, stride s): input (B,Hin,Win,Cin) -> output (B,Hout,Wout,Cout)
    const float* in;       // FWD / WGRAD: the convolution's input
    const float* w;        // FWD / DGRAD: [Cout][9][Cin]
    const float* bias;     // FWD
    const float* g;        // DGRAD / WGRAD: gradient w.r.t. the convolution's output, ALREADY masked by the ReLU behind that output (its producer did it)
    int B, Hin, Win, Cin, Hout, Wout, Cout, stride;
    // ---- the 1x1 stride-2 convolution on the block's shortcut, fused as extra K (FWD: of conv2; DGRAD: of conv1; WGRAD: extra tiles)
    const float* in2;      // FWD / WGRAD: the block's input (B,Hin2,Win2,Cin2)
    const float* w2;       // FWD / DGRAD: [Cout2][Cin2]
    const float* bias2;    // FWD
    const float* g2;       // DGRAD: (masked) gradient w.r.t. the block's output; pixels (B,H2,W2) with C2 channels
    int Hin2, Win2, Cin2;  // FWD / WGRAD: geometry of in2.  DGRAD: H2, W2 = output size of the shortcut conv, Cin2 = its Cout (K)
    // ---- epilogue
    const float* res;      // FWD: identity shortcut added before the ReLU.  DGRAD: (masked) gradient of the identity shortcut
    const float* outact;   // DGRAD: the forward activation the produced gradient belongs to: out = (acc + res) * (outact > 0)
    float* out;            // FWD: (B,Hout,Wout,Cout).  DGRAD: (B,Hin,Win,Cin).  WGRAD: gw [Cout][9][Cin]
    float* out2;           // WGRAD: gw2 [Cout][Cin2]
    float* gb;             // WGRAD: bias gradient [Cout] (NULL: none);  gb2: the shortcut conv's bias gradient (same sums)
    float* gb2;
    int relu, beta;        // FWD: ReLU on the output.  WGRAD: 1 = add to out / gb, 0 = overwrite
    // ---- decomposition
    int rows;              // GEMM rows: FWD B*Hout*Wout, DGRAD B*Hin*Win, WGRAD Cout
    int nchunk_main, nchunk;     // FWD / DGRAD: K chunks of the 3x3 part / in total.  WGRAD: pixel chunks
    int ntile_main;        // WGRAD: tiles of the 3x3 weights (the rest belong to w2)
    unsigned mg_hw, mg_w, mg_n8;        // ceil(2^32 / d) for d = (rows' H*W, W) and the padded tile count: n / d = umulhi(n, magic) for n, d < 2^16
    int zin, zin2, zg, zg2;             // offset (floats) of the workspace's 4 KB zero region from in / in2 / g / g2: where invalid pieces are read from
    float* partial; unsigned* ticket;   // cross-workgroup split (gridDim.y > 1)
    int stamp_slot;                     // ISLAM_POSE_STAMPS builds (scripts/debug/pose_head_stamps.py): which row of the phase-clock table this launch writes
};

#ifdef ISLAM_POSE_STAMPS                // phase clocks of one workgroup's wave 0 (never in the product build)
}  // namespace
__device__ long long islam_pose_stamps_buf[160 * 16];
namespace {
#define PSTAMP(j) do { __builtin_amdgcn_sched_barrier(0); if (stamp) { const long long now_ = clock64(); st_acc[j] += now_ - st_last; st_last = now_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSTAMP(j) do { } while (0)
#endif

// n / d for n * d < 2^32 with mg = ceil(2^32 / d): one multiply instead of the ~40-instruction integer division sequence (the row decode
// of a workgroup's set-up and of every weight-gradient chunk was 4-6 of them per thread)
__device__ __forceinline__ int fdiv(int n, unsigned mg) { return (int)__umulhi((unsigned)n, mg); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return float4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One workgroup = SIXTEEN wavefronts on one 32 x 32 output tile; K advances in chunks of KC = 128 and wave w multiplies k = 8w .. 8w+7 of
// every chunk (four MFMAs), so a SIMD holds four waves: while one issues MFMAs the others compute addresses, wait for loads and write
// LDS.  (The first version ran four waves per workgroup = one per SIMD: in-order issue serialised the address arithmetic, the load
// waits, the LDS traffic and the MFMAs of a chunk -- 5 800 clocks per chunk for 1 024 clocks of MFMA, scripts/debug/pose_head_stamps.py.)
// Every thread stages exactly one 16-byte piece of A and one of B per chunk.
//   FWD / DGRAD  A = gathered pixel rows, k contiguous: LDS [32 rows][KC + 1] (odd stride: conflict-free fragment reads).  k runs over
//                (tap, channel) of the 3x3 part padded to whole chunks, then over the shortcut convolution's channels.
//   FWD          B = weights [col][k], same layout.       DGRAD  B = weights [k = (tap, co)][32 ci].
//   WGRAD        A = (g*mask)[pixel][32 co], B = in[pixel@tap][32 ci], both [k = pixel][32].
constexpr int KC = 128, LDK = KC + 4;               // k-major LDS rows: 16-byte aligned, and (4 * row) mod 64 banks: conflict-free 16-byte fragment reads

// (the body of the kernels below: workgroup x of `ntiles` padded to a multiple of 8, split z of KS)
template <int MODE, int NW, bool HAS_DS>
__device__ __forceinline__ void conv_gemm_body(const ConvArgs& a, const int x, const int ntiles, const int z, const int KS, float* lds, int& s_last) {
    constexpr int T = 64 * NW;                         // threads
    constexpr int NP = 1024 / T;                       // 16-byte pieces of each operand a thread stages per chunk
    constexpr int A_FLOATS = (MODE == MODE_WGRAD) ? KC * 32 : 32 * LDK;
    constexpr int B_FLOATS = (MODE == MODE_FWD) ? 32 * LDK : KC * 32;
    constexpr int BUF = A_FLOATS + B_FLOATS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    // ---- which tile.  Workgroups are dealt to the eight XCDs round-robin; tile = a contiguous range per XCD, so that the workgroups
    // that share an operand (the column tiles of one pixel tile; the taps of one channel tile) fetch it through the same L2
    int tile;
    {
        const int xcd = x & 7, i = x >> 3, q = ntiles >> 3, r = ntiles & 7;
        if (i >= q + (xcd < r ? 1 : 0)) return;        // (padding of the tile count to a multiple of 8: the whole workgroup leaves)
        tile = xcd * q + (xcd < r ? xcd : r) + i;
    }
#ifdef ISLAM_POSE_STAMPS
    const bool stamp = tile == 0 && z == 0 && threadIdx.x == 0;
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = clock64();
    const long long st_wall0 = wall_clock64();
#endif
    int row0, col0;                                    // first GEMM row / column of the tile
    int tap_w = 0;                                     // WGRAD: the tap of this tile; ds: tile of the shortcut's weights
    bool ds_tile = false;
    if constexpr (MODE == MODE_WGRAD) {
        int tl = tile;
        ds_tile = tl >= a.ntile_main;
        if (ds_tile) tl -= a.ntile_main;
        const int nci = ds_tile ? a.Cin2 / 32 : (a.Cin == 4 ? 2 : a.Cin / 32);      // (first convolution: 36 = 9 taps x 4 channels in two column tiles)
        const int lci = 31 - __clz(nci), nco = a.Cout / 32, lco = 31 - __clz(nco);      // (channel counts are powers of two)
        const int ci_t = tl & (nci - 1);
        tl >>= lci;
        const int co_t = tl & (nco - 1);
        tap_w = tl >> lco;
        row0 = co_t * 32;
        col0 = ci_t * 32;
    } else {
        const int ncol = (MODE == MODE_FWD ? a.Cout : a.Cin) / 32;
        col0 = (tile & (ncol - 1)) * 32;
        row0 = (tile >> (31 - __clz(ncol))) * 32;
    }

    // ---- the pixel rows this thread stages (FWD / DGRAD), decoded ONCE: the pixel index its taps are offsets from and a 9-bit mask of the
    // taps that exist for it.  Per chunk that leaves a handful of integer instructions per piece: the waves of a SIMD share its issue
    // slots, and the address arithmetic -- not the MFMAs -- is what bounds these kernels (scripts/debug/pose_head_stamps.py).
    // Invalid pieces are not zeroed afterwards: their address is the ZERO TAIL every workspace buffer ends with (Plan::take).
    // piece j of a thread: k-major staging row srow + (T / 32) j, first k skq; row-major staging k row rrow + (T / 8) j, first column rnq
    const int srow = t >> 5, skq = (t & 31) * 4;
    const int rrow = t >> 3, rnq = (t & 7) * 4;
    int p0[NP];                                        // FWD: pixel of tap (0,0) in `in`;  DGRAD: pixel of tap (0,0) in g (taps SUBTRACT)
    unsigned valid9[NP];
    int p2[NP];                                        // pixel of the shortcut operand (FWD: (2y,2x) of in2; DGRAD: (y/2,x/2) of g2), -1: none
#pragma unroll
    for (int j = 0; j < NP; ++j) { p0[j] = 0; valid9[j] = 0; p2[j] = -1; }
    if constexpr (MODE != MODE_WGRAD) {
        const int H = MODE == MODE_FWD ? a.Hout : a.Hin, W = MODE == MODE_FWD ? a.Wout : a.Win;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int m = row0 + srow + (T / 32) * j;
            if (m < a.rows) {
                const int b = fdiv(m, a.mg_hw), r = m - b * (H * W), y = fdiv(r, a.mg_w), x = r - y * W;
                if constexpr (MODE == MODE_FWD) {
                    const int yb = y * a.stride - 1, xb = x * a.stride - 1;
                    p0[j] = (b * a.Hin + yb) * a.Win + xb;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int iy = yb + tap / 3, ix = xb + tap % 3;
                        valid9[j] |= (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) ? 1u << tap : 0u;
                    }
                    if (a.in2) p2[j] = (b * a.Hin2 + 2 * y) * a.Win2 + 2 * x;
                } else {
                    // input pixel (y, x) <- output pixel ((y + 1 - ky) / s, (x + 1 - kx) / s); for the taps that exist at stride 2 the
                    // quotient is ((y + 1) >> 1) - (ky >> 1)
                    const int yb = y + 1, xb = x + 1, sh = a.stride == 2 ? 1 : 0;
                    p0[j] = (b * a.Hout + (yb >> sh)) * a.Wout + (xb >> sh);
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        int ty = yb - tap / 3, tx = xb - tap % 3;
                        bool ok = ty >= 0 && tx >= 0;
                        if (sh) { ok = ok && !((ty | tx) & 1); ty >>= 1; tx >>= 1; }
                        valid9[j] |= (ok && ty < a.Hout && tx < a.Wout) ? 1u << tap : 0u;
                    }
                    if (a.g2 && !((y | x) & 1) && (y >> 1) < a.Hin2 && (x >> 1) < a.Win2) p2[j] = (b * a.Hin2 + (y >> 1)) * a.Win2 + (x >> 1);
                }
            }
        }
    }
    // invalid pieces read zeros from a 4 KB region, every thread its own 16 bytes (one address for all would be one cache line fetched by
    // every lane of every CU at once)
    const int zsp = (t & 255) * 4;
    const int lgK = 31 - __clz(MODE == MODE_FWD ? a.Cin : a.Cout);        // channels per tap of the K axis (a power of two)
    const int kmain = 9 << lgK, kmask = (1 << lgK) - 1;
    const int tap_sh = MODE == MODE_DGRAD && a.stride == 2 ? 1 : 0;
    const int tapW = MODE == MODE_FWD ? a.Win : a.Wout;

    const int c_begin = (int)((long long)a.nchunk * z / KS), c_end = (int)((long long)a.nchunk * (z + 1) / KS);
    // D register stages: while chunk c is multiplied the loads of chunks c+1 .. c+D are in flight
    constexpr int D = NP >= 4 ? 2 : 3;
    float4 ra_[D][NP], rb_[D][NP];

    // BRANCH-FREE by construction: with an `if` around the loads the compiler reuses the destination registers of a stage for address
    // arithmetic on the other path and guards that with s_waitcnt vmcnt(0) -- every chunk then waited for the loads it had just issued
    // (the first versions did: a full memory round trip per chunk, no overlap at all).  `live`: the chunk exists (c < c_end); chunks
    // past the end read the zero tail (their MFMAs add nothing).
    auto load = [&](int c, int st, bool live) {
        float4 (&ra)[NP] = ra_[st];
        float4 (&rb)[NP] = rb_[st];
        if constexpr (MODE == MODE_FWD) {
            const bool ds = HAS_DS && c >= a.nchunk_main;                  // (uniform) the shortcut's 1x1 stride-2 convolution: pixel (2y, 2x) of in2
            const int k = (ds ? c - a.nchunk_main : c) * KC + skq, tap = k >> lgK, ky = (tap * 11) >> 5, kx = tap - 3 * ky;
            const int dlt = ky * tapW + kx, kc = k & kmask;
            const float* pa = ds ? a.in2 : a.in;
            const float* pb = ds ? a.w2 : a.w;
            const int zt = (ds ? a.zin2 : a.zin) + zsp, ldb = ds ? a.Cin2 : kmain, kw = min(k, ldb - 4);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const bool okm = live && ((valid9[j] >> tap) & 1u);        // (tap >= 9 in the padded tail of the K axis: no bit)
                int o = okm ? ((p0[j] + dlt) << lgK) + kc : zt;
                if constexpr (HAS_DS) { const bool okd = live && k < a.Cin2 && p2[j] >= 0; o = ds ? (okd ? p2[j] * a.Cin2 + k : zt) : o; }
                ra[j] = ld4(pa + o);
                rb[j] = ld4(pb + (col0 + srow + (T / 32) * j) * ldb + kw);
            }
        } else if constexpr (MODE == MODE_DGRAD) {
            const bool ds = HAS_DS && c >= a.nchunk_main;                  // transposed shortcut: input pixel (y, x) <- output (y/2, x/2), both even
            const int cb = (ds ? c - a.nchunk_main : c) * KC;
            const int k = cb + skq, tap = k >> lgK, ky = (tap * 11) >> 5, kx = tap - 3 * ky;
            const int dlt = (ky >> tap_sh) * tapW + (kx >> tap_sh), kc = k & kmask;
            const float* pa = ds ? a.g2 : a.g;
            const float* pb = ds ? a.w2 : a.w;
            const int zt = (ds ? a.zg2 : a.zg) + zsp;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const bool okm = live && ((valid9[j] >> tap) & 1u);
                int o = okm ? ((p0[j] - dlt) << lgK) + kc : zt;
                if constexpr (HAS_DS) { const bool okd = live && k < a.Cin2 && p2[j] >= 0; o = ds ? (okd ? p2[j] * a.Cin2 + k : zt) : o; }
                ra[j] = ld4(pa + o);
                // B[k = (tap, co)][n = ci]: ci contiguous
                int kb = min(cb + rrow + (T / 8) * j, kmain - 1);
                int ob = ((kb & kmask) * 9 + (kb >> lgK)) * a.Cin;
                if constexpr (HAS_DS) { const int kd = min(cb + rrow + (T / 8) * j, a.Cin2 - 1); ob = ds ? kd * a.Cin : ob; }
                rb[j] = ld4(pb + ob + col0 + rnq);
            }
        } else {                                       // WGRAD: chunk c = output pixels [c * KC, (c + 1) * KC)
            const int npx = a.B * a.Hout * a.Wout;
            const bool first = a.Cin == 4;                                  // first convolution: one 16-byte piece = one tap's 4 channels
            const int tap = first ? col0 / 4 + (t & 7) : tap_w;
            const int ky = tap / 3, kx = tap - ky * 3;
            const float* pb = ds_tile ? a.in2 : a.in;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int m = c * KC + rrow + (T / 8) * j;
                const bool okm = live && m < npx;
                ra[j] = ld4(a.g + (okm ? m * a.Cout + row0 + rnq : a.zg + zsp));
                const int mm = okm ? m : 0;
                const int b = fdiv(mm, a.mg_hw), r = mm - b * (a.Hout * a.Wout), oy = fdiv(r, a.mg_w), ox = r - oy * a.Wout;
                const int iy = oy * a.stride - 1 + ky, ix = ox * a.stride - 1 + kx;
                const bool okb = okm && tap < 9 && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
                const int o_main = okb ? ((b * a.Hin + iy) * a.Win + ix) * a.Cin + (first ? 0 : col0 + rnq) : (first ? 0 : a.zin + zsp);
                const int o_ds = okm ? ((b * a.Hin2 + 2 * oy) * a.Win2 + 2 * ox) * a.Cin2 + col0 + rnq : a.zin2 + zsp;
                const float4 v = ld4(pb + (ds_tile ? o_ds : o_main));
                rb[j] = (first && !okb) ? zero4() : v;                      // (the network input has no zero tail)
            }
        }
    };

    auto stash = [&](int buf, int st) {
        float* sa = lds + buf * BUF;
        float* sb = sa + A_FLOATS;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if constexpr (MODE == MODE_WGRAD) {
                *reinterpret_cast<float4*>(sa + (rrow + (T / 8) * j) * 32 + rnq) = ra_[st][j];
                *reinterpret_cast<float4*>(sb + (rrow + (T / 8) * j) * 32 + rnq) = rb_[st][j];
            } else {
                *reinterpret_cast<float4*>(sa + (srow + (T / 32) * j) * LDK + skq) = ra_[st][j];
                if constexpr (MODE == MODE_FWD) *reinterpret_cast<float4*>(sb + (srow + (T / 32) * j) * LDK + skq) = rb_[st][j];
                else *reinterpret_cast<float4*>(sb + (rrow + (T / 8) * j) * 32 + rnq) = rb_[st][j];
            }
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float dbsum = 0.f;                                 // WGRAD: column sums of A (bias gradient), per lane
    const bool want_db = MODE == MODE_WGRAD && a.gb != nullptr && !ds_tile && tap_w == 0 && col0 == 0;

    // Wave w multiplies k = w KW .. w KW + KW - 1 of the chunk, eight at a time: MFMA (u, c) takes k = 8u + 4 (lane >> 5) + c from both operands
    // (any order of K is a valid order as long as A and B agree), so a k-major operand is read as ONE 16-byte fragment per four MFMAs.
    auto mma = [&](int buf) {
        const float* sa = lds + buf * BUF;
        const float* sb = sa + A_FLOATS;
        constexpr int KW = KC / NW, NU = KW / 8;
        const int ar = lane & 31, kb0 = wave * KW + 4 * (lane >> 5);
        float av[NU][4], bv[NU][4];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int k = kb0 + 8 * u;
            if constexpr (MODE == MODE_WGRAD) {
#pragma unroll
                for (int c = 0; c < 4; ++c) av[u][c] = sa[(k + c) * 32 + ar];
            } else {
                const float4 q = *reinterpret_cast<const float4*>(sa + ar * LDK + k);
                av[u][0] = q.x; av[u][1] = q.y; av[u][2] = q.z; av[u][3] = q.w;
            }
            if constexpr (MODE == MODE_FWD) {
                const float4 q = *reinterpret_cast<const float4*>(sb + ar * LDK + k);
                bv[u][0] = q.x; bv[u][1] = q.y; bv[u][2] = q.z; bv[u][3] = q.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) bv[u][c] = sb[(k + c) * 32 + ar];
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (MODE == MODE_WGRAD) dbsum += av[u][c];
#ifndef ISLAM_POSE_DBG_NO_MFMA
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][c], bv[u][c], acc, 0, 0, 0);
#else
                acc[c] += av[u][c] * bv[u][c];
#endif
            }
        }
    };

    PSTAMP(0);
    if (c_begin < c_end) {
#pragma unroll
        for (int st = 0; st < D; ++st) load(min(c_begin + st, c_end - 1), st, true);
        PSTAMP(1);
        stash(0, 0);
        __syncthreads();
        PSTAMP(2);
        int buf = 0;
        // no branch in the body but the exits: past the last chunk the prefetch re-reads the last chunk (never used)
        for (int c = c_begin; c < c_end; c += D) {
#pragma unroll
            for (int st = 0; st < D; ++st) {           // chunk c + st sits in LDS buffer `buf`; its register stage `st` is free again
                const int cc = c + st;
                if (cc >= c_end) goto chunks_done;
#ifndef ISLAM_POSE_DBG_NO_LOAD
                load(min(cc + D, c_end - 1), st, true);
#endif
                PSTAMP(3);
                mma(buf);
                PSTAMP(4);
                stash(buf ^ 1, (st + 1) % D);
                PSTAMP(5);
                __syncthreads();
                PSTAMP(6);
                buf ^= 1;
            }
        }
    chunks_done:;
    }

    // ---- the waves' partial sums are added through LDS in wave order; thread t owns elements e = t + T j (row e / 32, column e % 32)
    float v[NP], db = 0.f;
    {
        float* red = lds;                              // [wave][32][33]
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave * 1056 + ((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * 33 + (lane & 31)] = acc[i];
        float* dbr = lds + NW * 1056;                  // [wave][64]
        if constexpr (MODE == MODE_WGRAD) dbr[wave * 64 + lane] = dbsum;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int e = t + T * j, o = (e >> 5) * 33 + (e & 31);
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += red[w * 1056 + o];
            v[j] = sum;
        }
        if constexpr (MODE == MODE_WGRAD) {
            if (want_db && t < 32) {
#pragma unroll
                for (int w = 0; w < NW; ++w) db += dbr[w * 64 + t] + dbr[w * 64 + 32 + t];
            }
        }
    }

    // ---- cross-workgroup split: partial tiles to scratch (write-through stores), the last arriver adds them in split order
    if (KS > 1) {
        float* part = a.partial + ((size_t)tile * KS + z) * 1056;
#pragma unroll
        for (int j = 0; j < NP; ++j) st_agent(part + t + T * j, v[j]);
        if (want_db && t < 32) st_agent(part + 1024 + t, db);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) s_last = __hip_atomic_fetch_add(a.ticket + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(KS - 1);
        __syncthreads();
        if (!s_last) return;
        const float* q0 = a.partial + (size_t)tile * KS * 1056;
#pragma unroll
        for (int j = 0; j < NP; ++j) v[j] = 0.f;
        db = 0.f;
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int j = 0; j < NP; ++j) v[j] += ld_agent(q0 + (size_t)s * 1056 + t + T * j);
            if (want_db && t < 32) db += ld_agent(q0 + (size_t)s * 1056 + 1024 + t);
        }
        if (t == 0) __hip_atomic_store(a.ticket + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- epilogue: elements (row0 + e / 32, col0 + e % 32)
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int e = t + T * j, er = e >> 5, ec = e & 31;
        if constexpr (MODE == MODE_WGRAD) {
            const int col = col0 + ec;
            float* dst = ds_tile ? a.out2 + (size_t)(row0 + er) * a.Cin2 + col
                                 : (a.Cin == 4 ? a.out + (size_t)(row0 + er) * 36 + col : a.out + ((size_t)(row0 + er) * 9 + tap_w) * a.Cin + col);
            if (a.Cin != 4 || ds_tile || col < 36) *dst = a.beta ? *dst + v[j] : v[j];
        } else {
            const int m = row0 + er, col = col0 + ec;
            if (m < a.rows) {
                const int C = MODE == MODE_FWD ? a.Cout : a.Cin;
                const size_t o = (size_t)m * C + col;
                float x = v[j];
                if constexpr (MODE == MODE_FWD) {
                    x += a.bias[col];
                    if (a.bias2) x += a.bias2[col];
                    if (a.res) x += a.res[o];
                    if (a.relu) x = fmaxf(x, 0.f);
                } else {
                    if (a.res) x += a.res[o];                          // (the identity shortcut's gradient: already masked by its producer)
                    x = a.outact[o] > 0.f ? x : 0.f;                   // stored gradients carry the ReLU mask of the activation they belong to
                }
                a.out[o] = x;
            }
        }
    }
    if constexpr (MODE == MODE_WGRAD) {
        if (want_db && t < 32) {
            a.gb[row0 + t] = a.beta ? a.gb[row0 + t] + db : db;
            if (a.gb2) a.gb2[row0 + t] = a.beta ? a.gb2[row0 + t] + db : db;
        }
    }
#ifdef ISLAM_POSE_STAMPS
    PSTAMP(7);
    if (stamp) {
        long long* o = islam_pose_stamps_buf + (a.stamp_slot % 160) * 16;
        for (int q = 0; q < 8; ++q) o[q] = st_acc[q];
        o[8] = wall_clock64() - st_wall0;
        o[9] = c_end - c_begin; o[10] = MODE; o[11] = KC; o[12] = ntiles; o[13] = KS; o[14] = NW;
    }
#endif
}

// one convolution GEMM: 1-D grid of round_up(ntiles, 8) * KS workgroups
template <int MODE, int NW, bool HAS_DS>
__global__ __launch_bounds__(64 * NW, 1) void conv_gemm_kernel(const ConvArgs a, const int ntiles, const int KS) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // max(2 * BUF, NW * 1056 + NW * 64) floats; 16-byte fragments need the alignment
    __shared__ int s_last;
    const int n8 = (ntiles + 7) & ~7, z = fdiv(blockIdx.x, a.mg_n8);
    conv_gemm_body<MODE, NW, HAS_DS>(a, blockIdx.x - z * n8, ntiles, z, KS, lds, s_last);
}

// A data gradient and a weight gradient that read the SAME output gradient, in one launch (the backward of a convolution is this pair;
// neither needs the other): workgroups [0, n1) run the data gradient, the rest the weight gradient.  One dependent launch instead of
// two -- or of two streams joined by events, which cost ~6 us of queue gap per kernel (profiles/r06/pose_head_two_streams_r06.txt).
template <int NW, bool HAS_DS>
__global__ __launch_bounds__(64 * NW, 1) void conv_bwd_pair_kernel(const ConvArgs ad, const int ntiles_d, const int KS_d, const ConvArgs aw,
                                                                   const int ntiles_w, const int KS_w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_last;
    const int n8d = (ntiles_d + 7) & ~7, n1 = n8d * KS_d;
    if ((int)blockIdx.x < n1) {
        const int z = fdiv(blockIdx.x, ad.mg_n8);
        conv_gemm_body<MODE_DGRAD, NW, HAS_DS>(ad, blockIdx.x - z * n8d, ntiles_d, z, KS_d, lds, s_last);
    } else {
        const int id = blockIdx.x - n1, n8w = (ntiles_w + 7) & ~7, z = fdiv(id, aw.mg_n8);
        conv_gemm_body<MODE_WGRAD, NW, false>(aw, id - z * n8w, ntiles_w, z, KS_w, lds, s_last);
    }
}

// ------------------------------------------------------------------------------------------ the first convolution (4 -> 32, stride 2)
// K = 36: not worth a matrix-core tile.  Forward: one thread per (output pixel, 4 output channels), weights in LDS.
__global__ __launch_bounds__(TS) void stem0_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ y, int B, int Hin, int Win, int Hout, int Wout) {
    __shared__ float sw[32 * 36];
    for (int i = threadIdx.x; i < 32 * 36; i += TS) sw[i] = w[i];
    __syncthreads();
    const long long gid = (long long)blockIdx.x * TS + threadIdx.x;
    const int cg = (int)(gid & 7);
    const long long m = gid >> 3;
    if (m >= (long long)B * Hout * Wout) return;
    const int b = (int)(m / (Hout * Wout)), r = (int)(m - (long long)b * (Hout * Wout)), oy = r / Wout, ox = r - oy * Wout;
    float4 acc = ld4(bias + cg * 4);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = 2 * oy - 1 + tap / 3, ix = 2 * ox - 1 + tap % 3;
        if (iy < 0 || iy >= Hin || ix < 0 || ix >= Win) continue;
        const float4 v = ld4(x + ((size_t)(b * Hin + iy) * Win + ix) * 4);
        const float* wr = sw + cg * 4 * 36 + tap * 4;
        const float4 w0 = ld4(wr), w1 = ld4(wr + 36), w2 = ld4(wr + 72), w3 = ld4(wr + 108);
        acc.x = fmaf(v.x, w0.x, acc.x); acc.x = fmaf(v.y, w0.y, acc.x); acc.x = fmaf(v.z, w0.z, acc.x); acc.x = fmaf(v.w, w0.w, acc.x);
        acc.y = fmaf(v.x, w1.x, acc.y); acc.y = fmaf(v.y, w1.y, acc.y); acc.y = fmaf(v.z, w1.z, acc.y); acc.y = fmaf(v.w, w1.w, acc.y);
        acc.z = fmaf(v.x, w2.x, acc.z); acc.z = fmaf(v.y, w2.y, acc.z); acc.z = fmaf(v.z, w2.z, acc.z); acc.z = fmaf(v.w, w2.w, acc.z);
        acc.w = fmaf(v.x, w3.x, acc.w); acc.w = fmaf(v.y, w3.y, acc.w); acc.w = fmaf(v.z, w3.z, acc.w); acc.w = fmaf(v.w, w3.w, acc.w);
    }
    acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    *reinterpret_cast<float4*>(y + (size_t)m * 32 + cg * 4) = acc;
}

// ------------------------------------------------------------------------------------------ the two heads (Linear layers), B <= 16
// feat: (B, HW, 256) channels-last; the reference flattens NCHW (VOFlowNet.py:190): input index of the first Linear = c * HW + p.
constexpr int FC_IN = 1536, FC_H1 = 128, FC_H2 = 32, FC_MAXB = 16;

// h1[head][b][n] = relu(W1[head][n] . feat[b] + b1): one wave per (head, n); lanes walk the weight row in ITS order (coalesced), the
// features come from an LDS copy laid out in the same (NCHW-flatten) order
__global__ __launch_bounds__(TS) void fc1_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ w1t, const float* __restrict__ b1t,
                                                    const float* __restrict__ w1r, const float* __restrict__ b1r, float* __restrict__ h1, int B, int HW) {
    extern __shared__ __attribute__((aligned(16))) float sf[];                      // [B][FC_IN], column order of the weights
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, C = FC_IN / HW;
    for (int e = t; e < B * FC_IN; e += TS) {           // feat[b][p * C + c] -> sf[b][c * HW + p]
        const int b = e / FC_IN, i = e - b * FC_IN, p = i / C, c = i - p * C;
        sf[b * FC_IN + c * HW + p] = feat[e];
    }
    __syncthreads();
    const int o = blockIdx.x * 4 + wave, head = o / FC_H1, n = o - head * FC_H1;
    const float* w = (head ? w1r : w1t) + (size_t)n * FC_IN;
    float wv[FC_IN / 64];
#pragma unroll
    for (int j = 0; j < FC_IN / 64; ++j) wv[j] = w[lane + 64 * j];
    for (int b = 0; b < B; ++b) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < FC_IN / 64; ++j) s = fmaf(sf[b * FC_IN + lane + 64 * j], wv[j], s);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) h1[((size_t)head * B + b) * FC_H1 + n] = fmaxf(s + (head ? b1r : b1t)[n], 0.f);
    }
}

// one workgroup per head: h2 = relu(W2 h1 + b2), out[b][3 * head + j] = W3 h2 + b3
__global__ __launch_bounds__(TS) void fc23_fwd_kernel(const float* __restrict__ h1, const float* __restrict__ w2t, const float* __restrict__ b2t,
                                                     const float* __restrict__ w3t, const float* __restrict__ b3t, const float* __restrict__ w2r,
                                                     const float* __restrict__ b2r, const float* __restrict__ w3r, const float* __restrict__ b3r,
                                                     float* __restrict__ h2, float* __restrict__ out, int B) {
    __shared__ float s2[FC_MAXB * FC_H2];
    const int head = blockIdx.x, t = threadIdx.x;
    const float *w2 = head ? w2r : w2t, *b2 = head ? b2r : b2t, *w3 = head ? w3r : w3t, *b3 = head ? b3r : b3t;
    for (int o = t; o < B * FC_H2; o += TS) {
        const int b = o / FC_H2, n = o - b * FC_H2;
        float s = 0.f;
        for (int k = 0; k < FC_H1; ++k) s = fmaf(h1[((size_t)head * B + b) * FC_H1 + k], w2[n * FC_H1 + k], s);
        s = fmaxf(s + b2[n], 0.f);
        s2[o] = s;
        h2[(size_t)head * B * FC_H2 + o] = s;
    }
    __syncthreads();
    for (int o = t; o < B * 3; o += TS) {
        const int b = o / 3, j = o - b * 3;
        float s = 0.f;
        for (int k = 0; k < FC_H2; ++k) s = fmaf(s2[b * FC_H2 + k], w3[j * FC_H2 + k], s);
        out[b * 6 + head * 3 + j] = s + b3[j];
    }
}

// backward of the last two Linear layers, one workgroup per head: gW3, gb3, gW2, gb2 and g_h1 (masked by h1 > 0)
__global__ __launch_bounds__(TS) void fc23_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ h1, const float* __restrict__ h2,
                                                     const float* __restrict__ w2t, const float* __restrict__ w3t, const float* __restrict__ w2r,
                                                     const float* __restrict__ w3r, float* __restrict__ gw2t, float* __restrict__ gb2t,
                                                     float* __restrict__ gw3t, float* __restrict__ gb3t, float* __restrict__ gw2r,
                                                     float* __restrict__ gb2r, float* __restrict__ gw3r, float* __restrict__ gb3r,
                                                     float* __restrict__ gh1, int B, int beta) {
    __shared__ float sg[FC_MAXB * 3], sg2[FC_MAXB * FC_H2];
    const int head = blockIdx.x, t = threadIdx.x;
    const float *w2 = head ? w2r : w2t, *w3 = head ? w3r : w3t;
    float *gw2 = head ? gw2r : gw2t, *gb2 = head ? gb2r : gb2t, *gw3 = head ? gw3r : gw3t, *gb3 = head ? gb3r : gb3t;
    const float* h1h = h1 + (size_t)head * B * FC_H1;
    const float* h2h = h2 + (size_t)head * B * FC_H2;
    for (int o = t; o < B * 3; o += TS) sg[o] = gout[(o / 3) * 6 + head * 3 + o % 3];
    __syncthreads();
    for (int o = t; o < 3 * FC_H2; o += TS) {           // gW3[j][k] = sum_b g[b][j] h2[b][k]
        const int j = o / FC_H2, k = o - j * FC_H2;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(sg[b * 3 + j], h2h[b * FC_H2 + k], s);
        gw3[o] = beta ? gw3[o] + s : s;
    }
    if (t < 3) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += sg[b * 3 + t];
        gb3[t] = beta ? gb3[t] + s : s;
    }
    for (int o = t; o < B * FC_H2; o += TS) {           // g_h2 = (g W3) * (h2 > 0)
        const int b = o / FC_H2, k = o - b * FC_H2;
        float s = 0.f;
        for (int j = 0; j < 3; ++j) s = fmaf(sg[b * 3 + j], w3[j * FC_H2 + k], s);
        sg2[o] = h2h[o] > 0.f ? s : 0.f;
    }
    __syncthreads();
    for (int o = t; o < FC_H2 * FC_H1; o += TS) {       // gW2[n][k] = sum_b g_h2[b][n] h1[b][k]
        const int n = o / FC_H1, k = o - n * FC_H1;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(sg2[b * FC_H2 + n], h1h[b * FC_H1 + k], s);
        gw2[o] = beta ? gw2[o] + s : s;
    }
    if (t < FC_H2) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += sg2[b * FC_H2 + t];
        gb2[t] = beta ? gb2[t] + s : s;
    }
    for (int o = t; o < B * FC_H1; o += TS) {           // g_h1 = (g_h2 W2) * (h1 > 0)
        const int b = o / FC_H1, k = o - b * FC_H1;
        float s = 0.f;
        for (int n = 0; n < FC_H2; ++n) s = fmaf(sg2[b * FC_H2 + n], w2[n * FC_H1 + k], s);
        gh1[(size_t)head * B * FC_H1 + o] = h1h[o] > 0.f ? s : 0.f;
    }
}

// backward of the first Linear layers, weights: one workgroup per (head, n) row --
//   gW1[head][n][col] (+)= sum_b g_h1[head][b][n] * feat[b][i(col)],  col = c * HW + p, i = p * C + c;  gb1[head][n] (+)= sum_b g_h1[head][b][n]
__global__ __launch_bounds__(TS) void fc1_wgrad_kernel(const float* __restrict__ gh1, const float* __restrict__ feat, float* __restrict__ gw1t,
                                                      float* __restrict__ gb1t, float* __restrict__ gw1r, float* __restrict__ gb1r, int B, int HW,
                                                      int beta) {
    const int t = threadIdx.x, head = blockIdx.x / FC_H1, n = blockIdx.x - head * FC_H1, C = FC_IN / HW;
    float g[FC_MAXB];
#pragma unroll
    for (int b = 0; b < FC_MAXB; ++b) g[b] = b < B ? gh1[((size_t)head * B + b) * FC_H1 + n] : 0.f;
    float* gw = (head ? gw1r : gw1t) + (size_t)n * FC_IN;
#pragma unroll
    for (int j = 0; j < FC_IN / TS; ++j) {
        const int col = t + TS * j, c = col / HW, p = col - c * HW, i = p * C + c;
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < FC_MAXB; ++b)
            if (b < B) s = fmaf(g[b], feat[(size_t)b * FC_IN + i], s);
        gw[col] = beta ? gw[col] + s : s;
    }
    if (t == 0) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < FC_MAXB; ++b) s += g[b];
        float* gb = head ? gb1r : gb1t;
        gb[n] = beta ? gb[n] + s : s;
    }
}

// ... data: g_feat[b][i] = (feat[b][i] > 0) * sum_{head, n} g_h1[head][b][n] * W1[head][n][col(i)] (the gradient w.r.t. the last block's output).  Workgroup = 32 weight columns x 8 groups of 32 (head, n) rows; the eight partial sums are added in group order.
__global__ __launch_bounds__(TS) void fc1_dgrad_kernel(const float* __restrict__ gh1, const float* __restrict__ w1t, const float* __restrict__ w1r,
                                                      const float* __restrict__ feat, float* __restrict__ gfeat, int B, int HW) {
    __shared__ float sg[2 * FC_MAXB * FC_H1];          // g_h1 [head][b][n]
    __shared__ float sp[8 * FC_MAXB * 32];
    const int t = threadIdx.x, cl = t & 31, grp = t >> 5, col = blockIdx.x * 32 + cl, C = FC_IN / HW;
    for (int o = t; o < 2 * B * FC_H1; o += TS) sg[o] = gh1[o];
    float wv[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {                     // rows r = 32 * grp + j of the stacked [trans; rot] weight matrix
        const int r = 32 * grp + j, head = r / FC_H1, n = r - head * FC_H1;
        wv[j] = (head ? w1r : w1t)[(size_t)n * FC_IN + col];
    }
    __syncthreads();
    float acc[FC_MAXB];
#pragma unroll
    for (int b = 0; b < FC_MAXB; ++b) acc[b] = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int r = 32 * grp + j, head = r / FC_H1, n = r - head * FC_H1;
#pragma unroll
        for (int b = 0; b < FC_MAXB; ++b)
            if (b < B) acc[b] = fmaf(sg[(head * B + b) * FC_H1 + n], wv[j], acc[b]);
    }
#pragma unroll
    for (int b = 0; b < FC_MAXB; ++b) sp[(grp * FC_MAXB + b) * 32 + cl] = acc[b];
    __syncthreads();
    for (int o = t; o < B * 32; o += TS) {
        const int b = o / 32, c2 = o - b * 32, col2 = blockIdx.x * 32 + c2;
        float s = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 8; ++g2) s += sp[(g2 * FC_MAXB + b) * 32 + c2];
        const int c = col2 / HW, p = col2 - c * HW;
        const size_t fo = (size_t)b * FC_IN + p * C + c;
        gfeat[fo] = feat[fo] > 0.f ? s : 0.f;                // (stored gradients carry the ReLU mask of the activation they belong to)
    }
}

// ------------------------------------------------------------------------------------------ host side: the network as a list of layers
struct Geo { int H, W, C; };

struct Block {
    int cin, cout, stride;
    int p_conv1, p_conv2, p_ds;        // indices into the parameter table (weight; bias = +1); p_ds < 0: identity shortcut
    Geo gin, gout;
    size_t a_h, a_out;                 // workspace offsets (floats) of conv1's and the block's output
    size_t g_h, g_out;                 // ... of the gradients w.r.t. them (every gradient has a buffer of its own: the weight-gradient
                                       // kernels read them from a second stream while the data-gradient chain moves on)
};

constexpr int N_PARAMS = 120;
constexpr int STAGES[5][2] = {{64, 3}, {128, 4}, {128, 6}, {256, 7}, {256, 3}};

struct Plan {
    int B, H, W;
    Geo g_stem_geo;                    // 32 channels at H/2 x W/2
    size_t a_stem[3];
    std::vector<Block> blocks;
    size_t a_zero, a_h1, a_h2, a_gh1, g_stem[3], a_partial[2], a_ticket[2], floats, partial_floats;      // scratch set 0: main stream, 1: weight-gradient stream
    int HWf;
};

inline int half_up(int v) { return (v - 1) / 2 + 1; }      // 3x3 / stride 2 / pad 1 (and 1x1 / stride 2): floor((v - 1) / 2) + 1

constexpr size_t PARTIAL_FLOATS = (size_t)1 << 21;            // 8 MB: max over the launches of tiles * splits * 1056 (checked per launch)
constexpr int N_TICKETS = 4096;

bool make_plan(int B, int H, int W, Plan& p) {
    p.B = B; p.H = H; p.W = W;
    size_t off = 0;
    auto take = [&](size_t n) { const size_t o = off; off += (n + 63) / 64 * 64; return o; };
    p.a_zero = take(1024);                                   // nobody writes it: the workspace is zero-filled once, kernels read invalid pieces from here
    p.g_stem_geo = Geo{half_up(H), half_up(W), 32};
    const size_t stem_n = (size_t)B * p.g_stem_geo.H * p.g_stem_geo.W * 32;
    for (int i = 0; i < 3; ++i) p.a_stem[i] = take(stem_n);
    Geo g = p.g_stem_geo;
    int pi = 6;
    p.blocks.clear();
    for (int s = 0; s < 5; ++s) {
        for (int b = 0; b < STAGES[s][1]; ++b) {
            Block k;
            k.cin = g.C; k.cout = STAGES[s][0]; k.stride = b == 0 ? 2 : 1;
            k.gin = g;
            k.gout = b == 0 ? Geo{half_up(g.H), half_up(g.W), k.cout} : Geo{g.H, g.W, k.cout};
            k.p_conv1 = pi; k.p_conv2 = pi + 2; pi += 4;
            if (b == 0) { k.p_ds = pi; pi += 2; } else k.p_ds = -1;
            const size_t n = (size_t)B * k.gout.H * k.gout.W * k.cout;
            k.a_h = take(n);
            k.a_out = take(n);
            k.g_h = take(n);
            k.g_out = take(n);
            p.blocks.push_back(k);
            g = k.gout;
        }
    }
    if (pi != 108) return false;
    p.HWf = g.H * g.W;
    if (g.C * p.HWf != FC_IN || B > FC_MAXB || B < 1) return false;
    p.a_h1 = take((size_t)2 * B * FC_H1);
    p.a_h2 = take((size_t)2 * B * FC_H2);
    p.a_gh1 = take((size_t)2 * B * FC_H1);
    for (int i = 0; i < 3; ++i) p.g_stem[i] = take(stem_n);
    p.partial_floats = PARTIAL_FLOATS;
    for (int i = 0; i < 2; ++i) {
        p.a_partial[i] = take(p.partial_floats);
        p.a_ticket[i] = take(N_TICKETS);
    }
    p.floats = off;
    return true;
}

inline int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return dev;
}

template <int MODE>
constexpr size_t lds_bytes(int nw) {
    const int a_floats = (MODE == MODE_WGRAD) ? KC * 32 : 32 * LDK, b_floats = (MODE == MODE_FWD) ? 32 * LDK : KC * 32;
    const int red = nw * 1056 + nw * 64, two = 2 * (a_floats + b_floats);
    return sizeof(float) * (size_t)(two > red ? two : red);
}

// Cross-workgroup split of K: only when the tiles alone leave most of the chip idle -- a split costs a write-through round trip, an
// atomic and a read-back across XCDs (~5 us, scripts/debug/pose_head_stamps.py) on top of the launch.
int pick_split(int ntiles, int nchunk) {
    // (A/B knob) ISLAM_POSE_SPLIT_BIG=k: layers with 128 .. 400 tiles and >= 8 chunks split k ways, so that every CU holds two or more
    // workgroups of equal length instead of one long one (or two on some CUs, one on others)
    static const int big = [] { const char* e = std::getenv("ISLAM_POSE_SPLIT_BIG"); const int v = e ? std::atoi(e) : 1; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
    if (ntiles >= 128) return (ntiles <= 400 && nchunk >= 8) ? big : 1;
    if (nchunk < 4) return 1;
    int ks = 1;
    while (ks < nchunk / 2 && ntiles * ks < 192) ++ks;
    return ks;
}

#ifdef ISLAM_POSE_STAMPS
static int g_stamp_slot = 0;
#endif

inline bool pow2(int v) { return v > 0 && !(v & (v - 1)); }

struct Prepared { ConvArgs a; int ntiles, KS; };

// fills the decomposition fields of `a`; kdim / kdim2: channels per tap of the K axis (FWD: Cin / Cin2, DGRAD: Cout / Cout of the shortcut)
template <int MODE>
int prepare(ConvArgs a, int kdim, int kdim2, float* partial, size_t partial_floats, unsigned* ticket, Prepared& out) {
#ifdef ISLAM_POSE_STAMPS
    a.stamp_slot = g_stamp_slot++;
#endif
    int ntiles;
    if constexpr (MODE == MODE_WGRAD) {
        const int npx = a.B * a.Hout * a.Wout;
        if ((a.Cin != 4 && a.Cin % 32) || a.Cout % 32 || (a.in2 && a.Cin2 % 32)) return fail(ISLAM_EARG, "pose head: channel counts must be multiples of 32");
        a.rows = a.Cout;
        a.nchunk_main = a.nchunk = (npx + KC - 1) / KC;
        a.ntile_main = a.Cin == 4 ? (a.Cout / 32) * 2 : (a.Cout / 32) * (a.Cin / 32) * 9;
        ntiles = a.ntile_main + (a.in2 ? (a.Cout / 32) * (a.Cin2 / 32) : 0);
    } else {
        // the K axis: (tap, channel) of the 3x3 part padded to whole chunks, then the shortcut convolution's channels likewise
        if (!pow2(kdim) || kdim % 32 || (kdim2 && kdim2 % 4)) return fail(ISLAM_EARG, "pose head: %d / %d channels (a power of two >= 32 is needed)", kdim, kdim2);
        const int ncol = (MODE == MODE_FWD ? a.Cout : a.Cin) / 32;
        a.nchunk_main = (9 * kdim + KC - 1) / KC;
        a.nchunk = a.nchunk_main + (kdim2 ? (kdim2 + KC - 1) / KC : 0);
        ntiles = ((a.rows + 31) / 32) * ncol;
    }
    int KS = pick_split(ntiles, a.nchunk);
    while (KS > 1 && (size_t)ntiles * KS * 1056 > partial_floats) --KS;
    if (ntiles > N_TICKETS) KS = 1;
    a.partial = partial;
    a.ticket = ticket;
    {
        auto magic = [](int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
        const int H = MODE == MODE_DGRAD ? a.Hin : a.Hout, W = MODE == MODE_DGRAD ? a.Win : a.Wout, n8 = (ntiles + 7) & ~7;
        // (n / d == umulhi(n, ceil(2^32 / d)) for n * d < 2^32)
        if ((long long)a.B * H * W * H * W >= (1ll << 32) || (long long)n8 * KS * n8 >= (1ll << 32)) return fail(ISLAM_EARG, "pose head: %d x %d x %d pixels exceed the row decode", a.B, H, W);
        a.mg_hw = magic(H * W); a.mg_w = magic(W); a.mg_n8 = magic(n8);
    }
    out.a = a; out.ntiles = ntiles; out.KS = KS;
    return ISLAM_OK;
}

// wavefronts per workgroup (ISLAM_POSE_NW = 4 / 8 / 16 for A/B runs): 8 = two per SIMD -- one wave's MFMAs run beside the other's address
// arithmetic and LDS traffic; 16 halves the MFMAs per staged piece and is issue-bound, 4 serialises the two
int waves_per_wg() {
    static const int nw = [] { const char* e = std::getenv("ISLAM_POSE_NW"); const int v = e ? std::atoi(e) : 8; return v == 4 || v == 16 ? v : 8; }();
    return nw;
}

template <int MODE, int NW, bool HAS_DS>
int launch_single_ds(const Prepared& p, hipStream_t s) {
    constexpr size_t lds = lds_bytes<MODE>(NW);
    static bool attr_set[64] = {};                           // per device: the attribute lives in the device's code object
    const int dev = current_device();
    if (!attr_set[dev]) {
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)conv_gemm_kernel<MODE, NW, HAS_DS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((conv_gemm_kernel<MODE, NW, HAS_DS>), dim3(((p.ntiles + 7) & ~7) * p.KS), dim3(64 * NW), lds, s, p.a, p.ntiles, p.KS);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}
template <int MODE, int NW>
int launch_single(const Prepared& p, hipStream_t s) {
    if constexpr (MODE == MODE_WGRAD) return launch_single_ds<MODE, NW, false>(p, s);
    else return p.a.nchunk > p.a.nchunk_main ? launch_single_ds<MODE, NW, true>(p, s) : launch_single_ds<MODE, NW, false>(p, s);
}

template <int MODE>
int launch_conv(const ConvArgs& a, int kdim, int kdim2, hipStream_t s, float* partial, size_t partial_floats, unsigned* ticket) {
    Prepared p;
    if (int rc = prepare<MODE>(a, kdim, kdim2, partial, partial_floats, ticket, p)) return rc;
    const int nw = waves_per_wg();
    if (nw == 4) return launch_single<MODE, 4>(p, s);
    if (nw == 16) return launch_single<MODE, 16>(p, s);
    return launch_single<MODE, 8>(p, s);
}

template <int NW, bool HAS_DS>
int launch_pair_ds(const Prepared& d, const Prepared& w, hipStream_t s) {
    constexpr size_t lds = lds_bytes<MODE_DGRAD>(NW) > lds_bytes<MODE_WGRAD>(NW) ? lds_bytes<MODE_DGRAD>(NW) : lds_bytes<MODE_WGRAD>(NW);
    static bool attr_set[64] = {};
    const int dev = current_device();
    if (!attr_set[dev]) {
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)conv_bwd_pair_kernel<NW, HAS_DS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    const int grid = ((d.ntiles + 7) & ~7) * d.KS + ((w.ntiles + 7) & ~7) * w.KS;
    hipLaunchKernelGGL((conv_bwd_pair_kernel<NW, HAS_DS>), dim3(grid), dim3(64 * NW), lds, s, d.a, d.ntiles, d.KS, w.a, w.ntiles, w.KS);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}
template <int NW>
int launch_pair_nw(const Prepared& d, const Prepared& w, hipStream_t s) {
    return d.a.nchunk > d.a.nchunk_main ? launch_pair_ds<NW, true>(d, w, s) : launch_pair_ds<NW, false>(d, w, s);
}

// data gradient `ad` and weight gradient `aw` of one convolution in one launch (separate split-K scratch sets)
int launch_bwd_pair(const ConvArgs& ad, int kdim, int kdim2, const ConvArgs& aw, hipStream_t s, float* const partial[2], size_t partial_floats,
                    unsigned* const ticket[2]) {
    Prepared d, w;
    if (int rc = prepare<MODE_DGRAD>(ad, kdim, kdim2, partial[0], partial_floats, ticket[0], d)) return rc;
    if (int rc = prepare<MODE_WGRAD>(aw, 0, 0, partial[1], partial_floats, ticket[1], w)) return rc;
    const int nw = waves_per_wg();
    if (nw == 4) return launch_pair_nw<4>(d, w, s);
    if (nw == 16) return launch_pair_nw<16>(d, w, s);
    return launch_pair_nw<8>(d, w, s);
}

// Fork / join of a second stream around the caller's, with events (valid in eager execution and under stream capture alike).
struct Fork {
    hipStream_t main = nullptr, side = nullptr;
    size_t next = 0;
    static hipStream_t side_stream() {
        static hipStream_t st = nullptr;
        if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
        return st;
    }
    static std::vector<hipEvent_t>& pool() { static std::vector<hipEvent_t> ev; return ev; }
    hipEvent_t event() {
        auto& ev = pool();
        if (next == ev.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            ev.push_back(e);
        }
        return ev[next++];
    }
    int begin(hipStream_t s) {
        main = s;
        const char* one = std::getenv("ISLAM_POSE_ONE_STREAM");
        if (one && one[0] == '1') { side = s; return ISLAM_OK; }       // (s may be the null stream)
        side = side_stream();
        if (!side) return fail(ISLAM_EHIP, "pose head: cannot create the weight-gradient stream");
        return ISLAM_OK;
    }
    int sync_side() {                                  // the side stream waits for everything enqueued on the main stream so far
        if (side == main) return ISLAM_OK;
        hipEvent_t e = event();
        if (!e) return fail(ISLAM_EHIP, "pose head: cannot create an event");
        ISLAM_HIP_CHECK(hipEventRecord(e, main));
        ISLAM_HIP_CHECK(hipStreamWaitEvent(side, e, 0));
        return ISLAM_OK;
    }
    int join() {                                       // the main stream waits for the side stream
        if (side == main) return ISLAM_OK;
        hipEvent_t e = event();
        if (!e) return fail(ISLAM_EHIP, "pose head: cannot create an event");
        ISLAM_HIP_CHECK(hipEventRecord(e, side));
        ISLAM_HIP_CHECK(hipStreamWaitEvent(main, e, 0));
        return ISLAM_OK;
    }
};

int check_common(const char* fn, const void* x, const void* const* params, const void* ws, size_t ws_bytes, int B, int H, int W, Plan& p) {
    if (!x || !params || !ws) return fail(ISLAM_EARG, "%s: null pointer", fn);
    if (!make_plan(B, H, W, p)) return fail(ISLAM_EARG, "%s: B=%d (1..%d) and a %dx%d input whose 256-channel feature map is not 6 pixels", fn, B, FC_MAXB, H, W);
    if (ws_bytes < p.floats * sizeof(float)) return fail(ISLAM_EARG, "%s: workspace of %zu bytes, %zu needed", fn, ws_bytes, p.floats * sizeof(float));
    for (int i = 0; i < N_PARAMS; ++i)
        if (!params[i]) return fail(ISLAM_EARG, "%s: parameter %d is null", fn, i);
    return ISLAM_OK;
}

}  // namespace

#ifdef ISLAM_POSE_STAMPS
extern "C" int islam_pose_stamps(long long* out, int reset_slot) {
    if (reset_slot >= 0) g_stamp_slot = reset_slot;
    return out ? (hipMemcpyFromSymbol(out, HIP_SYMBOL(islam_pose_stamps_buf), sizeof(long long) * 160 * 16) == hipSuccess ? 0 : 1) : 0;
}
#endif

extern "C" {

size_t islam_pose_head_workspace_bytes(int B, int H, int W) {
    Plan p;
    if (!make_plan(B, H, W, p)) return 0;
    return p.floats * sizeof(float);
}

int islam_pose_head_forward(const float* x, const float* const* params, float* out6, void* workspace, size_t workspace_bytes, int B, int H, int W,
                            void* stream) {
    Plan p;
    if (int rc = check_common("islam_pose_head_forward", x, (const void* const*)params, workspace, workspace_bytes, B, H, W, p)) return rc;
    if (!out6) return fail(ISLAM_EARG, "islam_pose_head_forward: null output");
    hipStream_t s = as_stream(stream);
    float* ws = (float*)workspace;
    float* partial = ws + p.a_partial[0];
    unsigned* ticket = (unsigned*)(ws + p.a_ticket[0]);
    const Geo gs = p.g_stem_geo;
    {
        const long long threads = (long long)B * gs.H * gs.W * 8;
        hipLaunchKernelGGL(stem0_fwd_kernel, dim3((unsigned)((threads + TS - 1) / TS)), dim3(TS), 0, s, x, params[0], params[1], ws + p.a_stem[0], B, H, W,
                           gs.H, gs.W);
        ISLAM_LAUNCH_CHECK();
    }
    auto conv = [&](const float* in, Geo gi, int pw, float* out, Geo go, int stride, const float* res, const float* in2, Geo gi2, int pw2) {
        ConvArgs a;
        std::memset(&a, 0, sizeof(a));
        a.in = in; a.w = params[pw]; a.bias = params[pw + 1];
        a.B = B; a.Hin = gi.H; a.Win = gi.W; a.Cin = gi.C; a.Hout = go.H; a.Wout = go.W; a.Cout = go.C; a.stride = stride;
        if (in2) { a.in2 = in2; a.w2 = params[pw2]; a.bias2 = params[pw2 + 1]; a.Hin2 = gi2.H; a.Win2 = gi2.W; a.Cin2 = gi2.C; }
        a.res = res; a.out = out; a.relu = 1;
        a.rows = B * go.H * go.W;
        a.zin = (int)((ws + p.a_zero) - in);
        a.zin2 = in2 ? (int)((ws + p.a_zero) - in2) : 0;
        return launch_conv<MODE_FWD>(a, gi.C, in2 ? gi2.C : 0, s, partial, p.partial_floats, ticket);
    };
    if (int rc = conv(ws + p.a_stem[0], gs, 2, ws + p.a_stem[1], gs, 1, nullptr, nullptr, Geo{}, 0)) return rc;
    if (int rc = conv(ws + p.a_stem[1], gs, 4, ws + p.a_stem[2], gs, 1, nullptr, nullptr, Geo{}, 0)) return rc;
    const float* cur = ws + p.a_stem[2];
    for (const Block& k : p.blocks) {
        float* h = ws + k.a_h;
        float* o = ws + k.a_out;
        if (int rc = conv(cur, k.gin, k.p_conv1, h, k.gout, k.stride, nullptr, nullptr, Geo{}, 0)) return rc;
        if (k.p_ds >= 0) {
            if (int rc = conv(h, k.gout, k.p_conv2, o, k.gout, 1, nullptr, cur, k.gin, k.p_ds)) return rc;
        } else {
            if (int rc = conv(h, k.gout, k.p_conv2, o, k.gout, 1, cur, nullptr, Geo{}, 0)) return rc;
        }
        cur = o;
    }
    float* h1 = ws + p.a_h1;
    float* h2 = ws + p.a_h2;
    {
        static bool attr_set[64] = {};
        const int dev = current_device();
        if (!attr_set[dev]) {
            ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)fc1_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FC_MAXB * FC_IN * sizeof(float))));
            attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL(fc1_fwd_kernel, dim3(2 * FC_H1 / 4), dim3(TS), (size_t)B * FC_IN * sizeof(float), s, cur, params[108], params[109], params[114],
                       params[115], h1, B, p.HWf);
    ISLAM_LAUNCH_CHECK();
    hipLaunchKernelGGL(fc23_fwd_kernel, dim3(2), dim3(TS), 0, s, h1, params[110], params[111], params[112], params[113], params[116], params[117],
                       params[118], params[119], h2, out6, B);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

int islam_pose_head_backward(const float* x, const float* const* params, float* const* grads, const float* grad_out6, void* workspace,
                             size_t workspace_bytes, int B, int H, int W, int accumulate, void* stream) {
    Plan p;
    if (int rc = check_common("islam_pose_head_backward", x, (const void* const*)params, workspace, workspace_bytes, B, H, W, p)) return rc;
    if (!grads || !grad_out6) return fail(ISLAM_EARG, "islam_pose_head_backward: null pointer");
    for (int i = 0; i < N_PARAMS; ++i)
        if (!grads[i]) return fail(ISLAM_EARG, "islam_pose_head_backward: gradient %d is null", i);
    hipStream_t s = as_stream(stream);
    float* ws = (float*)workspace;
    const int beta = accumulate ? 1 : 0;
    const Block& last = p.blocks.back();
    const float* feat = ws + last.a_out;
    float* h1 = ws + p.a_h1;
    float* h2 = ws + p.a_h2;
    float* gh1 = ws + p.a_gh1;
    // The backward of a convolution = a data gradient and a weight gradient that read the same output gradient and do not need each
    // other: they go out as ONE launch (conv_bwd_pair_kernel), so the chain of dependent launches is as long as the forward's.
    // ISLAM_POSE_TWO_STREAMS=1 (A/B runs): separate launches, the weight gradients on a second stream forked / joined with events.
    static const bool two_streams = [] { const char* e = std::getenv("ISLAM_POSE_TWO_STREAMS"); return e && e[0] == '1'; }();
    Fork fk;
    hipStream_t sw = s;
    if (two_streams) {
        if (int rc = fk.begin(s)) return rc;
        sw = fk.side;
    }
    float* partial[2] = {ws + p.a_partial[0], ws + p.a_partial[1]};
    unsigned* ticket[2] = {(unsigned*)(ws + p.a_ticket[0]), (unsigned*)(ws + p.a_ticket[1])};
    // ---- heads
    hipLaunchKernelGGL(fc23_bwd_kernel, dim3(2), dim3(TS), 0, s, grad_out6, h1, h2, params[110], params[112], params[116], params[118], grads[110],
                       grads[111], grads[112], grads[113], grads[116], grads[117], grads[118], grads[119], gh1, B, beta);
    ISLAM_LAUNCH_CHECK();
    if (two_streams) { if (int rc = fk.sync_side()) return rc; }
    hipLaunchKernelGGL(fc1_wgrad_kernel, dim3(2 * FC_H1), dim3(TS), 0, sw, gh1, feat, grads[108], grads[109], grads[114], grads[115], B, p.HWf, beta);
    ISLAM_LAUNCH_CHECK();
    hipLaunchKernelGGL(fc1_dgrad_kernel, dim3(FC_IN / 32), dim3(TS), 0, s, gh1, params[108], params[114], feat, ws + last.g_out, B, p.HWf);
    ISLAM_LAUNCH_CHECK();
    // ---- feature net, last block first.  g_out = gradient w.r.t. the block's output (not yet masked by its ReLU)
    // Every stored gradient already carries the ReLU mask of the activation it belongs to (its producer's epilogue applies it), so the
    // kernels read plain fp32 operands.
    auto wargs = [&](const float* in, Geo gi, const float* g, Geo go, int stride, int pw, const float* in2, Geo gi2, int pw2) {
        ConvArgs a;
        std::memset(&a, 0, sizeof(a));
        a.in = in; a.g = g;
        a.B = B; a.Hin = gi.H; a.Win = gi.W; a.Cin = gi.C; a.Hout = go.H; a.Wout = go.W; a.Cout = go.C; a.stride = stride;
        a.out = grads[pw]; a.gb = grads[pw + 1]; a.beta = beta;
        a.zin = gi.C == 4 ? 0 : (int)((ws + p.a_zero) - in);      // (the network input is not part of the workspace: its invalid pieces are zeroed after the load)
        a.zg = (int)((ws + p.a_zero) - g);
        a.zin2 = in2 ? (int)((ws + p.a_zero) - in2) : 0;
        if (in2) { a.in2 = in2; a.Hin2 = gi2.H; a.Win2 = gi2.W; a.Cin2 = gi2.C; a.out2 = grads[pw2]; a.gb2 = grads[pw2 + 1]; }
        return a;
    };
    // data gradient of the convolution with weights pw: g (B, go) -> out (B, gi), masked by outact (the activation `out` is the gradient of)
    auto dargs = [&](const float* g, Geo go, int pw, int stride, float* out, const float* outact, Geo gi, const float* res, const float* g2, Geo go2,
                     int pw2) {
        ConvArgs a;
        std::memset(&a, 0, sizeof(a));
        a.g = g; a.w = params[pw];
        a.B = B; a.Hin = gi.H; a.Win = gi.W; a.Cin = gi.C; a.Hout = go.H; a.Wout = go.W; a.Cout = go.C; a.stride = stride;
        if (g2) { a.g2 = g2; a.w2 = params[pw2]; a.Hin2 = go2.H; a.Win2 = go2.W; a.Cin2 = go2.C; }
        a.res = res; a.outact = outact; a.out = out;
        a.rows = B * gi.H * gi.W;
        a.zg = (int)((ws + p.a_zero) - g);
        a.zg2 = g2 ? (int)((ws + p.a_zero) - g2) : 0;
        return a;
    };
    // backward of one convolution: its weight gradient `aw` and the data gradient `ad` (kd / kd2: channels per tap of ad's K axis)
    auto backward_of = [&](const ConvArgs& ad, int kd, int kd2, const ConvArgs& aw) -> int {
        if (!two_streams) return launch_bwd_pair(ad, kd, kd2, aw, s, partial, p.partial_floats, ticket);
        if (int rc = fk.sync_side()) return rc;            // the gradient both read was produced by the last launch on the main stream
        if (int rc = launch_conv<MODE_WGRAD>(aw, 0, 0, sw, partial[1], p.partial_floats, ticket[1])) return rc;
        return launch_conv<MODE_DGRAD>(ad, kd, kd2, s, partial[0], p.partial_floats, ticket[0]);
    };
    for (int bi = (int)p.blocks.size() - 1; bi >= 0; --bi) {
        const Block& k = p.blocks[bi];
        const float* xin = bi == 0 ? ws + p.a_stem[2] : ws + p.blocks[bi - 1].a_out;
        float* g_x = bi == 0 ? ws + p.g_stem[2] : ws + p.blocks[bi - 1].g_out;
        const float* h = ws + k.a_h;
        float* g2 = ws + k.g_out;                          // = d loss / d out * (out > 0)
        float* g_h = ws + k.g_h;
        // conv2 (and the shortcut convolution): weight gradients from g2, and g_h = conv2^T(g2) * (h > 0)
        if (int rc = backward_of(dargs(g2, k.gout, k.p_conv2, 1, g_h, h, k.gout, nullptr, nullptr, Geo{}, 0), k.gout.C, 0,
                                 wargs(h, k.gout, g2, k.gout, 1, k.p_conv2, k.p_ds >= 0 ? xin : nullptr, k.gin, k.p_ds))) return rc;
        // conv1: weight gradient from g_h, and g_x = (conv1^T(g_h) + shortcut) * (x > 0): identity -> + g2; convolution -> + ds^T(g2) as extra K
        const ConvArgs aw1 = wargs(xin, k.gin, g_h, k.gout, k.stride, k.p_conv1, nullptr, Geo{}, 0);
        if (k.p_ds >= 0) {
            if (int rc = backward_of(dargs(g_h, k.gout, k.p_conv1, k.stride, g_x, xin, k.gin, nullptr, g2, k.gout, k.p_ds), k.gout.C, k.gout.C, aw1)) return rc;
        } else {
            if (int rc = backward_of(dargs(g_h, k.gout, k.p_conv1, 1, g_x, xin, k.gin, g2, nullptr, Geo{}, 0), k.gout.C, 0, aw1)) return rc;
        }
    }
    // ---- the three plain convolutions in front (conv + ReLU)
    const Geo gs = p.g_stem_geo;
    {
        float* g_2 = ws + p.g_stem[2];
        float* g_1 = ws + p.g_stem[1];
        float* g_0 = ws + p.g_stem[0];
        if (int rc = backward_of(dargs(g_2, gs, 4, 1, g_1, ws + p.a_stem[1], gs, nullptr, nullptr, Geo{}, 0), 32, 0,
                                 wargs(ws + p.a_stem[1], gs, g_2, gs, 1, 4, nullptr, Geo{}, 0))) return rc;
        if (int rc = backward_of(dargs(g_1, gs, 2, 1, g_0, ws + p.a_stem[0], gs, nullptr, nullptr, Geo{}, 0), 32, 0,
                                 wargs(ws + p.a_stem[0], gs, g_1, gs, 1, 2, nullptr, Geo{}, 0))) return rc;
        // the first convolution (4 -> 32, stride 2): weight gradient only; the same kernel, its 36 = 9 taps x 4 channels as two column tiles
        if (two_streams) { if (int rc = fk.sync_side()) return rc; }
        if (int rc = launch_conv<MODE_WGRAD>(wargs(x, Geo{H, W, 4}, g_0, gs, 2, 0, nullptr, Geo{}, 0), 0, 0, sw, partial[1], p.partial_floats, ticket[1])) return rc;
    }
    if (two_streams) { if (int rc = fk.join()) return rc; }
    return ISLAM_OK;
}

}  // extern "C"
