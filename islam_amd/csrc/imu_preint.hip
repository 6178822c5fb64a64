// IMU pre-integration on gfx950, float64 / float32.
//
// Replaces (reference file:line under /root/reference):
//   imu_integrator.py:116-158  the per-frame Python loop of IMUModule.integrate
//                              (each iteration ~60 tiny kernels + 3 D2H syncs)
//   pp.module.IMUPreintegrator.forward = integrate + predict (PyPose, external);
//   its covariance propagation is discarded by the reference and is not computed.
//
// Floating-point contract: the results are defined to be bit-identical to the plain-C restatement
// in oracle/imu_preint.c.  Every operation is one IEEE operation in the working precision in the
// order written there (FMA contraction is disabled for this translation unit), the doubling scan of
// pp.cumprod keeps its Hillis-Steele association order, cumsum and the frame-to-frame state chain
// are sequential, sqrt/divide are correctly rounded, and sin/cos come from the fdlibm k_sin/k_cos
// polynomials evaluated in double.
//
// Kernels (data stays in HBM/L2 between them; sizes are tiny, the path is launch-latency bound):
//   A  scan_kernel   one wave per frame: dr_j = Exp(gyro_j dt_j), doubling scan in LDS -> incre_r
//   B  chain_rot_kernel  four lanes (one quaternion component each): R0_{i+1} = R0_i * incre_r_i[F]   (strictly sequential, like the reference)
//   C  frame_kernel  one lane per frame: a_j, cumsum of dv / dp / dt, rotate by R0_i
//   D  (world mode, inside chain_kernel's second pass) p/v chain
#include <hip/hip_runtime.h>

#include "common.h"

#pragma clang fp contract(off)

using namespace islam;

namespace {

__device__ __forceinline__ double ksin(double x) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x + v * (S1 + z * r);
}
__device__ __forceinline__ double kcos(double x) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double ax = fabs(x);
    if (ax < 0.3) return 1.0 - (0.5 * z - z * r);
    double qx = (ax > 0.78125) ? 0.28125 : 0.25 * ax;
    double hz = 0.5 * z - qx;
    double a = 1.0 - qx;
    return a - (hz - z * r);
}
__device__ __forceinline__ void sincos_contract(double x, double* s, double* c) {
    const double pio4 = 7.85398163397448278999e-01, invpio2 = 6.36619772367581382433e-01,
                 pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    double y = x;
    int q = 0;
    if (fabs(x) > pio4) {
        double fn = rint(x * invpio2);
        y = (x - fn * pio2_1) - fn * pio2_1t;
        q = (int)((long long)fn & 3);
    }
    double sy = ksin(y), cy = kcos(y);
    switch (q) {
        case 0: *s = sy; *c = cy; break;
        case 1: *s = cy; *c = -sy; break;
        case 2: *s = -sy; *c = -cy; break;
        default: *s = -cy; *c = sy; break;
    }
}

template <class T> struct Eps;
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };
template <> struct Eps<float> { static constexpr float v = 1.1920928955078125e-07f; };

template <class T> __device__ __forceinline__ T sqrt_rn(T x);
template <> __device__ __forceinline__ double sqrt_rn<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float sqrt_rn<float>(float x) { return sqrtf(x); }

template <class T> struct Q { T x, y, z, w; };

template <class T> __device__ __forceinline__ Q<T> qmul(Q<T> a, Q<T> b) {
    Q<T> o;
    o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    o.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    o.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    return o;
}
template <class T> __device__ __forceinline__ void qact(Q<T> q, const T* p, T* o) {
    T cx = (T)2 * (q.y * p[2] - q.z * p[1]);
    T cy = (T)2 * (q.z * p[0] - q.x * p[2]);
    T cz = (T)2 * (q.x * p[1] - q.y * p[0]);
    T ox = p[0] + q.w * cx + (q.y * cz - q.z * cy);
    T oy = p[1] + q.w * cy + (q.z * cx - q.x * cz);
    T oz = p[2] + q.w * cz + (q.x * cy - q.y * cx);
    o[0] = ox; o[1] = oy; o[2] = oz;
}
template <class T> __device__ __forceinline__ Q<T> so3exp(T px, T py, T pz) {
    T th2 = px * px + py * py + pz * pz;
    T th = sqrt_rn<T>(th2);
    T imag, real;
    if (th > Eps<T>::v) {
        double s, c;
        sincos_contract((double)((T)0.5 * th), &s, &c);
        imag = (T)s / th;
        real = (T)c;
    } else {
        T th4 = th2 * th2;
        imag = (T)0.5 - (T)(1.0 / 48.0) * th2 + (T)(1.0 / 3840.0) * th4;
        real = (T)1.0 - (T)(1.0 / 8.0) * th2 + (T)(1.0 / 384.0) * th4;
    }
    return {px * imag, py * imag, pz * imag, real};
}
template <class T> __device__ __forceinline__ Q<T> ldq(const T* p) { return {p[0], p[1], p[2], p[3]}; }
template <class T> __device__ __forceinline__ void stq(Q<T> q, T* p) { p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w; }

// A: per frame i, incre_r (F_i + 1 quaternions) -> ir[(seg[i] + i) .. ]   (frame i owns F_i + 1 slots)
template <class T>
__global__ __launch_bounds__(64) void scan_kernel(const T* __restrict__ dt, const T* __restrict__ gyro,
                                                   const int64_t* __restrict__ seg, T* __restrict__ ir, int* __restrict__ zero_word) {
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) { zero_word[0] = 0; zero_word[1] = 0; }   // chain_world_kernel's published-rows counter + its gave-up flag
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* buf0 = reinterpret_cast<T*>(smem_raw);
    const int i = blockIdx.x;
    const int a = (int)seg[i], F = (int)(seg[i + 1] - seg[i]);
    const int L = F + 1;
    T* buf1 = buf0 + 4 * L;
    for (int j = threadIdx.x; j < L; j += 64) {
        Q<T> q{0, 0, 0, 1};
        if (j > 0) {
            const int sidx = a + j - 1;
            const T d = dt[sidx];
            q = so3exp<T>(gyro[3 * sidx] * d, gyro[3 * sidx + 1] * d, gyro[3 * sidx + 2] * d);
        }
        stq(q, buf0 + 4 * j);
    }
    __syncthreads();
    T* cur = buf0;
    T* nxt = buf1;
    for (int s = 1; s < L; s *= 2) {             // pp.cumprod(left=False): x[i] <- x[i-s] * x[i] for all i >= s at once
        for (int j = threadIdx.x; j < L; j += 64) {
            Q<T> v = ldq(cur + 4 * j);
            if (j >= s) v = qmul(ldq(cur + 4 * (j - s)), v);
            stq(v, nxt + 4 * j);
        }
        __syncthreads();
        T* t = cur; cur = nxt; nxt = t;
    }
    T* o = ir + 4 * ((size_t)a + i);
    for (int j = threadIdx.x; j < L; j += 64) stq(ldq(cur + 4 * j), o + 4 * j);
}

// B: sequential rotation chain over frames.  R0[i] = rotation at the start of frame i, R0[nframes] = final.
// The products must be taken strictly left to right (bit-exact contract: the reference chains `rot = rot * drot` frame by frame,
// imu_integrator.py:151), so the chain itself is serial -- but a quaternion product is four independent sums of four products, and
// in every one of them term k multiplies the SAME component of the running rotation (qmul: a.w, a.x, a.y, a.z in that order).  Four
// lanes of a wavefront walk the chain together, one output component each: per frame a lane reads its four signed factors
// +-b[sigma_c(k)] from a table the rest of the workgroup has staged in LDS (negation is exact, so a - x y == a + x (-y) bit for bit),
// takes the running rotation's component k from the lane that holds it with a DPP quad broadcast (a VALU move, no LDS round trip),
// and does 4 multiplies + 3 adds in the order qmul writes them.  One lane alone spent ~330 clocks per frame (28 dependent double
// operations issued at wave rate + LDS round trips for operands and results: 0.78 ms for 5000 frames); the quad form ~70.
constexpr int CHAIN_CHUNK = 256;      // frames per staged chunk of the p / v chain (7 terms per frame), two buffers
constexpr int ROT_CHUNK = 128;        // frames per staged chunk of the rotation chain (16 factors per frame), two buffers
static_assert(CHAIN_CHUNK % 64 == 0 && ROT_CHUNK % 64 == 0, "a chunk is whole wavefronts of frames (ballot masks) and an even number of register blocks");

// double <-> two dwords through a DPP quad permutation (ctrl = p0 | p1 << 2 | p2 << 4 | p3 << 6: lane i of a quad reads lane p_i)
template <int CTRL> __device__ __forceinline__ double quad_perm_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);      // (every lane of a full quad has a source: `old` is never taken;
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);      //  passing the source itself saves the move that zeroes it)
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ float quad_perm_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ double quad_bcast(double v) { return quad_perm_d<CTRL>(v); }
template <int CTRL> __device__ __forceinline__ float quad_bcast(float v) { return quad_perm_f<CTRL>(v); }

// One frame of the rotation chain on the four-lane form: the lane's component of qmul(r, b) from its four signed factors.
// double: each product is ONE v_fmac_f64 whose first operand is a DPP row broadcast of the component it needs (gfx90a+ "DP ALU" DPP:
// row_newbcast only; v_mul_f64 / v_add_f64 are VOP3 and take no DPP operand, the VOP2 v_fmac_f64 does) onto an accumulator preset
// to -0.0: fma(a, f, -0.0) IS round(a f), sign of a zero product included (-0 + x == x for every x) -- instead of two 32-bit DPP
// moves in front of every multiply on the chain.  The builtin has no 64-bit form, hence the assembler text; the four presets come
// first, which also gives the DPP reads their two wait states after the VALU write of r (the hazard recogniser does not look
// into inline assembly).  The sums are taken in the order qmul writes them.
__device__ __forceinline__ double rot_step(double rc, double f0, double f1, double f2, double f3) {
    double m0, m1, m2, m3;
    const double nz = -0.0;
    asm volatile(
        "v_mov_b64 %0, %9\n\tv_mov_b64 %1, %9\n\tv_mov_b64 %2, %9\n\tv_mov_b64 %3, %9\n\t"
        "v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %6 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf"
        : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3)
        : "v"(rc), "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(nz));
    return m0 + m1 + m2 + m3;
}
__device__ __forceinline__ float rot_step(float rc, float f0, float f1, float f2, float f3) {
    const float aw = quad_bcast<0xff>(rc), ax = quad_bcast<0x00>(rc), ay = quad_bcast<0x55>(rc), az = quad_bcast<0xaa>(rc);
    return aw * f0 + ax * f1 + ay * f2 + az * f3;
}

constexpr int ROT_U = 8;              // frames per register block of the walk: one LDS round trip per block instead of per frame

// rows of R0 that every agent may read: published (release, agent scope) by the rotation chain's workgroup after each chunk it has
// copied out, awaited (acquire) by the p / v chain's workgroup of chain_world_kernel
__device__ __forceinline__ void publish_rows(int* ready, int rows) {
    __threadfence();
    __hip_atomic_store(ready, rows, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// The wait is BOUNDED (as trial_elim_kernel's ticket wait is): if the producer workgroup never publishes -- it faulted, or another call
// sharing this scratch re-zeroed the counter (islam_hip.h: concurrent calls need their own scratch) -- the waiting lane gives up after
// AWAIT_LIMIT_TICKS of the constant-rate 100 MHz wall clock (0.5 s whatever the shader clock and however many persistent kernels of
// other streams compete for the producer's CU), raises the sticky flag ready[1], and chain_world_kernel poisons the p / v rows with
// NaN instead of hanging the stream without a diagnostic.
constexpr long long AWAIT_LIMIT_TICKS = 50000000LL;
__device__ __forceinline__ void await_rows(int* ready, int rows) {
    if ((threadIdx.x & 63) == 0 && __hip_atomic_load(ready + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < rows) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > AWAIT_LIMIT_TICKS) { __hip_atomic_store(ready + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __threadfence();
}

template <class T>
struct RotLds {
    T sb[2][ROT_CHUNK + 3 * ROT_U][4][4];
    T out[2][ROT_CHUNK + 2 * ROT_U][4];      // R0[base + 1 + j]
    unsigned long long hasm[2][ROT_CHUNK / 64 + 1];                        // bit j: F_j > 0 (frames past the chunk: 0)
};

// ready != nullptr: the number of valid rows of R0 is published after every chunk (chain_world_kernel)
template <class T>
__device__ __forceinline__ void chain_rot_body(RotLds<T>& L, const int64_t* __restrict__ seg, int nframes, const T* __restrict__ ir,
                                               const T* __restrict__ init_rot, T* __restrict__ R0, T* __restrict__ rot_copy, int* ready) {
    // sb[.][j][c][k]: factor of term k of output component c (x, y, z, w) for frame j's increment b = incre_r_j[F_j], in the order
    // qmul multiplies the running rotation's components w, x, y, z:
    //   x: a.w b.x + a.x b.w + a.y b.z - a.z b.y    y: a.w b.y - a.x b.z + a.y b.w + a.z b.x
    //   z: a.w b.z + a.x b.y - a.y b.x + a.z b.w    w: a.w b.w - a.x b.x - a.y b.y - a.z b.z
    // Two buffers: while wave 0 walks chunk k, waves 1-3 stage chunk k + 1 (two dependent global round trips per frame: offsets,
    // then the increment) and copy chunk k - 1 out -- the walk is all that is left on the kernel's critical path.
    // (rows past a chunk's last frame are read by the walk's look-ahead and never used)
    auto& sb = L.sb;
    auto& out = L.out;
    auto& hasm = L.hasm;
    const int tid = threadIdx.x;
    const int nchunk = (nframes + ROT_CHUNK - 1) / ROT_CHUNK;
    // lane c < 4 of wave 0 carries component c of the running rotation (x, y, z, w)
    T rc = tid < 4 ? init_rot[tid] : (T)0;
    if (tid < 4) { R0[tid] = rc; if (rot_copy) rot_copy[tid] = rc; }
    if (tid < 2) hasm[tid][ROT_CHUNK / 64] = 0;
    auto stage = [&](int k, int t, int nt) {                 // threads t = 0 .. nt - 1 (whole wavefronts) fill buffer k & 1 with chunk k
        const int base = k * ROT_CHUNK, cnt = min(ROT_CHUNK, nframes - base), b = k & 1;
        for (int j = t; j < ROT_CHUNK; j += nt) {
            bool h = false;
            if (j < cnt) {
                const int i = base + j;
                const int a = (int)seg[i], F = (int)(seg[i + 1] - seg[i]);
                h = F > 0;
                const T* src = ir + 4 * ((size_t)a + i + F);
                const T bx = src[0], by = src[1], bz = src[2], bw = src[3];
                T* o = &sb[b][j][0][0];
                o[0] = bx;  o[1] = bw;   o[2] = bz;   o[3] = -by;
                o[4] = by;  o[5] = -bz;  o[6] = bw;   o[7] = bx;
                o[8] = bz;  o[9] = by;   o[10] = -bx; o[11] = bw;
                o[12] = bw; o[13] = -bx; o[14] = -by; o[15] = -bz;
            }
            const unsigned long long m = __ballot(h);
            if ((t & 63) == 0) hasm[b][j >> 6] = m;
        }
    };
    auto copy_out = [&](int k, int t, int nt) {
        const int base = k * ROT_CHUNK, cnt = min(ROT_CHUNK, nframes - base);
        const T* o = &out[k & 1][0][0];
        for (int j = t; j < cnt * 4; j += nt) {
            const T v = o[j];
            R0[4 * (size_t)(base + 1) + j] = v;
            if (rot_copy) rot_copy[4 * (size_t)(base + 1) + j] = v;
        }
    };
    stage(0, tid, 256);
    __syncthreads();
    for (int k = 0; k < nchunk; ++k) {
        if (tid >= 64) {
            if (k + 1 < nchunk) stage(k + 1, tid - 64, 192);
            if (k > 0) copy_out(k - 1, tid - 64, 192);
        } else if (tid < 4) {
            // Blocks of ROT_U frames: the block's factors sit in registers (requested one block ahead, 16-byte LDS reads at
            // immediate offsets), so the walk itself is 4 multiplies + 3 adds + one LDS store per frame and its dependency chain
            // (multiply -> three adds) is all that is left of a frame.  A frame without IMU samples keeps the rotation
            // (imu_integrator.py:134-140): a uniform branch on the chunk's bit mask, no select on the chain.
            const int cnt = min(ROT_CHUNK, nframes - k * ROT_CHUNK);
            const T* fb = &sb[k & 1][0][tid][0];
            T* ob = &out[k & 1][0][tid];
            const unsigned char* hb = reinterpret_cast<const unsigned char*>(&hasm[k & 1][0]);
            T fa[ROT_U][4], fn[ROT_U][4];
            auto load = [&](T (&f)[ROT_U][4], int& hv, int blk) {       // (the block's mask byte travels with its factors: one block ahead)
#pragma unroll
                for (int u = 0; u < ROT_U; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) f[u][q] = fb[(size_t)(blk * ROT_U + u) * 16 + q];
                hv = hb[blk];
            };
            auto walk = [&](const T (&f)[ROT_U][4], int hv, int blk) {
                const unsigned bits = __builtin_amdgcn_readfirstlane(hv);
                T* o = ob + (size_t)blk * ROT_U * 4;
                if (__builtin_expect(bits == 0xffu, 1)) {       // every frame of the block has samples: no branch per frame
#pragma unroll
                    for (int u = 0; u < ROT_U; ++u) {
                        rc = rot_step(rc, f[u][0], f[u][1], f[u][2], f[u][3]);      // = qmul(r, b) component c
                        o[u * 4] = rc;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < ROT_U; ++u) {
                        if (bits & (1u << u)) rc = rot_step(rc, f[u][0], f[u][1], f[u][2], f[u][3]);
                        o[u * 4] = rc;
                    }
                }
            };
            asm volatile("s_nop 4" ::: "memory");              // (EXEC was narrowed to four lanes just above: DPP wait states)
            const int nblk = (cnt + ROT_U - 1) / ROT_U;
            int ha, hn;
            load(fa, ha, 0);
            for (int blk = 0; blk < nblk; blk += 2) {
                load(fn, hn, blk + 1);
                walk(fa, ha, blk);
                load(fa, ha, blk + 2);
                walk(fn, hn, blk + 1);
            }
        }
        __syncthreads();
        if (ready && tid == 0) publish_rows(ready, 1 + k * ROT_CHUNK);          // R0[0 .. k ROT_CHUNK] are in memory
    }
    copy_out(nchunk - 1, tid, 256);
    if (ready) {
        __syncthreads();
        if (tid == 0) publish_rows(ready, nframes + 1);
    }
}

template <class T>
__global__ __launch_bounds__(256) void chain_rot_kernel(const int64_t* __restrict__ seg, int nframes, const T* __restrict__ ir,
                                                         const T* __restrict__ init_rot, T* __restrict__ R0, T* __restrict__ rot_copy) {
    __shared__ __attribute__((aligned(16))) RotLds<T> L;
    chain_rot_body<T>(L, seg, nframes, ir, init_rot, R0, rot_copy, nullptr);
}

// C: per frame (one lane each): local integration, rotated into the frame's start orientation.
//    loc[i] = { R0_i incre_v[F] (3), R0_i incre_p[F] (3), incre_t (1) }
template <class T>
__device__ __forceinline__ void frame_loc(int i, const T* __restrict__ dt, const T* __restrict__ acc, const int64_t* __restrict__ seg,
                                          const T* __restrict__ ir, const T* __restrict__ R0, T gravity, T (&o)[7]) {
    const int a = (int)seg[i], F = (int)(seg[i + 1] - seg[i]);
    const Q<T> r0 = ldq(R0 + 4 * (size_t)i);
    const T* irf = ir + 4 * ((size_t)a + i);
    const T g[3] = {0, 0, gravity};
    T iv[3] = {0, 0, 0}, ip[3] = {0, 0, 0}, it = 0;
    for (int j = 0; j < F; ++j) {
        Q<T> q = qmul(r0, ldq(irf + 4 * (j + 1)));
        Q<T> qi{-q.x, -q.y, -q.z, q.w};
        T gb[3], av[3], ra[3];
        qact(qi, g, gb);
        av[0] = acc[3 * (a + j)] - gb[0];
        av[1] = acc[3 * (a + j) + 1] - gb[1];
        av[2] = acc[3 * (a + j) + 2] - gb[2];
        qact(ldq(irf + 4 * j), av, ra);
        const T d = dt[a + j];
        const T d2 = d * d;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            T dp = iv[c] * d + ra[c] * (T)0.5 * d2;
            ip[c] = ip[c] + dp;
            iv[c] = iv[c] + ra[c] * d;
        }
        it = it + d;
    }
    T rv[3], rp[3];
    qact(r0, iv, rv);
    qact(r0, ip, rp);
    o[0] = rv[0]; o[1] = rv[1]; o[2] = rv[2]; o[3] = rp[0]; o[4] = rp[1]; o[5] = rp[2]; o[6] = it;
}

template <class T>
__global__ __launch_bounds__(64) void frame_kernel(const T* __restrict__ dt, const T* __restrict__ acc,
                                                    const int64_t* __restrict__ seg, int nframes,
                                                    const T* __restrict__ ir, const T* __restrict__ R0, T gravity,
                                                    T* __restrict__ loc) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nframes) return;
    T o[7];
    frame_loc<T>(i, dt, acc, seg, ir, R0, gravity, o);
    T* d = loc + 7 * (size_t)i;
#pragma unroll
    for (int c = 0; c < 7; ++c) d[c] = o[c];
}

template <class T>
struct ChainLds {
    T sl[2][CHAIN_CHUNK + 3 * ROT_U][7];      // (rows past a chunk's last frame: look-ahead only)
    T so[2][CHAIN_CHUNK + 2 * ROT_U][6];
    unsigned long long hasm[2][CHAIN_CHUNK / 64 + 1];                       // bit j: F_j > 0
};
// what the p / v chain needs to compute its frame terms itself (chain_world_kernel); ready == nullptr: they are read from `loc`
template <class T>
struct FrameSrc { int* ready; const T* dt; const T* acc; const T* ir; T gravity; T* loc_out; };

template <class T>
__device__ __forceinline__ void chain_pv_body(ChainLds<T>& L, const int64_t* __restrict__ seg, int nframes, const T* __restrict__ R0,
                                              const T* __restrict__ loc, const T* __restrict__ init_pos, const T* __restrict__ init_vel,
                                              T* __restrict__ out_pos, T* __restrict__ out_vel, const FrameSrc<T>& fr) {
    // world mode: sequential p/v chain (three lanes of wave 0, one per coordinate).  Two LDS buffers: while wave 0 walks chunk k,
    // waves 1-3 stage the per-frame terms of chunk k + 1 (all of a thread's global loads issued before the first LDS store: one
    // memory round trip per chunk instead of ten) and copy chunk k - 1 out.  (The rotations of the world rows are written by
    // chain_rot_kernel.)
    auto& sl = L.sl;
    auto& so = L.so;
    auto& hasm = L.hasm;
    const int tid = threadIdx.x;
    const int nchunk = (nframes + CHAIN_CHUNK - 1) / CHAIN_CHUNK;
    if (tid < 3) { out_pos[tid] = init_pos[tid]; out_vel[tid] = init_vel[tid]; }
    if (tid < 2) hasm[tid][CHAIN_CHUNK / 64] = 0;
    constexpr int SL_ITEMS = (CHAIN_CHUNK * 7 + 191) / 192;          // loc values per staging thread (192 threads in the loop)
    auto stage = [&](int k, int t, int nt) {
        const int base = k * CHAIN_CHUNK, cnt = min(CHAIN_CHUNK, nframes - base), b = k & 1;
        if (fr.ready) {
            // chain_world_kernel: the frame terms are computed HERE (frame_kernel's arithmetic, one frame per thread) as soon as the
            // rotation chain's workgroup has published the chunk's rows of R0 -- the chain of this workgroup runs a chunk or two behind
            // the other one instead of a whole kernel behind it
            await_rows(fr.ready, base + cnt);
            for (int j = t; j < CHAIN_CHUNK; j += nt) {
                bool h = false;
                if (j < cnt) {
                    T o[7];
                    frame_loc<T>(base + j, fr.dt, fr.acc, seg, fr.ir, R0, fr.gravity, o);
                    h = seg[base + j + 1] > seg[base + j];
#pragma unroll
                    for (int c = 0; c < 7; ++c) sl[b][j][c] = o[c];
                    if (fr.loc_out) {
                        T* d = fr.loc_out + 7 * (size_t)(base + j);
#pragma unroll
                        for (int c = 0; c < 7; ++c) d[c] = o[c];
                    }
                }
                const unsigned long long m = __ballot(h);
                if ((t & 63) == 0) hasm[b][j >> 6] = m;
            }
            return;
        }
        T r[SL_ITEMS];
        const T* src = loc + 7 * (size_t)base;
#pragma unroll
        for (int i = 0; i < SL_ITEMS; ++i) { const int e = t + i * nt; r[i] = e < cnt * 7 ? src[e] : (T)0; }
        for (int j = t; j < CHAIN_CHUNK; j += nt) {
            const bool h = j < cnt && seg[base + j + 1] > seg[base + j];
            const unsigned long long m = __ballot(h);
            if ((t & 63) == 0) hasm[b][j >> 6] = m;
        }
        T* dst = &sl[b][0][0];
#pragma unroll
        for (int i = 0; i < SL_ITEMS; ++i) { const int e = t + i * nt; if (e < cnt * 7) dst[e] = r[i]; }
    };
    auto copy_out = [&](int k, int t, int nt) {
        const int base = k * CHAIN_CHUNK, cnt = min(CHAIN_CHUNK, nframes - base);
        for (int j = t; j < cnt * 3; j += nt) {
            const int f = j / 3, c = j - 3 * f;
            out_pos[3 * (size_t)(base + 1) + j] = so[k & 1][f][c];
            out_vel[3 * (size_t)(base + 1) + j] = so[k & 1][f][3 + c];
        }
    };
    if (tid < 192) stage(0, tid, 192);                   // (the prologue uses the loop's 192-thread mapping: whole wavefronts)
    __syncthreads();
    T p = 0, v = 0, sp = 0;
    if (tid < 3) { p = init_pos[tid]; v = init_vel[tid]; sp = p; }
    for (int k = 0; k < nchunk; ++k) {
        if (tid >= 64) {
            if (k + 1 < nchunk) stage(k + 1, tid - 64, 192);
            if (k > 0) copy_out(k - 1, tid - 64, 192);
        } else if (tid < 3) {
            // the three coordinates are independent chains: lane c walks coordinate c.  Blocks of ROT_U frames with the block's terms
            // in registers, requested one block ahead (a frame-by-frame walk waits for an LDS round trip per frame); a frame without
            // samples (imu_integrator.py:134-140: vel zeroed, pos / rot held) is a uniform branch on the chunk's bit mask
            const int c = tid, b = k & 1;
            const int cnt = min(CHAIN_CHUNK, nframes - k * CHAIN_CHUNK);
            const unsigned char* hb = reinterpret_cast<const unsigned char*>(&hasm[b][0]);
            T fa[ROT_U][3], fn[ROT_U][3];
            auto load = [&](T (&f)[ROT_U][3], int& hv, int blk) {
#pragma unroll
                for (int u = 0; u < ROT_U; ++u) {
                    const T* r = &sl[b][blk * ROT_U + u][0];
                    f[u][0] = r[c]; f[u][1] = r[3 + c]; f[u][2] = r[6];
                }
                hv = hb[blk];
            };
            auto walk = [&](const T (&f)[ROT_U][3], int hv, int blk) {
                const unsigned bits = __builtin_amdgcn_readfirstlane(hv);
                T* o = &so[b][blk * ROT_U][c];
                if (__builtin_expect(bits == 0xffu, 1)) {       // every frame of the block has samples: no branch per frame
#pragma unroll
                    for (int u = 0; u < ROT_U; ++u) {
                        const T sv = v + f[u][0];
                        sp = p + f[u][1] + v * f[u][2];
                        o[u * 6] = sp; o[u * 6 + 3] = sv; p = sp; v = sv;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < ROT_U; ++u) {
                        T sv = 0;
                        if (bits & (1u << u)) {
                            asm volatile("" ::: "memory");      // (keeps the branch: as selects, two v_cndmask sit on the p chain of every frame)
                            sv = v + f[u][0];
                            sp = p + f[u][1] + v * f[u][2];
                        }
                        o[u * 6] = sp; o[u * 6 + 3] = sv; p = sp; v = sv;
                    }
                }
            };
            const int nblk = (cnt + ROT_U - 1) / ROT_U;
            int ha, hn;
            load(fa, ha, 0);
            for (int blk = 0; blk < nblk; blk += 2) {
                load(fn, hn, blk + 1);
                walk(fa, ha, blk);
                load(fa, ha, blk + 2);
                walk(fn, hn, blk + 1);
            }
            // (only the LAST chunk has blocks past its frames -- a full chunk is an even number of blocks -- so the velocity those
            // zero is never carried on)
        }
        __syncthreads();
    }
    copy_out(nchunk - 1, tid, 256);
    if (fr.ready) {                                      // a wait gave up (await_rows): no silent garbage
        __syncthreads();
        if (__hip_atomic_load(fr.ready + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            const T nan = (T)__builtin_nan("");
            for (int j = tid; j < 3 * (nframes + 1); j += 256) { out_pos[j] = nan; out_vel[j] = nan; }
        }
    }
}

// D: outputs.  world mode: sequential p/v chain (row 0 = init).  motion mode: every frame starts from p = v = 0.
// (launched with 64 or 256 threads; without the bound the compiler budgets registers for 1024-thread blocks -- 128 per lane -- and the
// world-mode walk spilled 17 of them)
template <class T>
__global__ __launch_bounds__(256) void finish_kernel(const int64_t* __restrict__ seg, int nframes, const T* __restrict__ R0,
                              const T* __restrict__ loc, const T* __restrict__ init_pos, const T* __restrict__ init_vel,
                              int motion_mode, T* __restrict__ out_pos, T* __restrict__ out_rot, T* __restrict__ out_vel) {
    if (motion_mode) {
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= nframes) return;
        const int F = (int)(seg[i + 1] - seg[i]);
        const T* l = loc + 7 * (size_t)i;
        const T zero = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // predict with p0 = v0 = 0:  vel = 0 + R0 iv ; pos = (0 + R0 ip) + 0 * t
            out_vel[3 * (size_t)i + c] = (F > 0) ? (zero + l[c]) : zero;
            out_pos[3 * (size_t)i + c] = (F > 0) ? ((zero + l[3 + c]) + zero * l[6]) : zero;
        }
        Q<T> a = ldq(R0 + 4 * (size_t)i), b = ldq(R0 + 4 * (size_t)(i + 1));
        Q<T> ai{-a.x, -a.y, -a.z, a.w};
        stq(qmul(ai, b), out_rot + 4 * (size_t)i);
        return;
    }
    if (blockIdx.x != 0) return;
    __shared__ __attribute__((aligned(16))) ChainLds<T> L;
    const FrameSrc<T> none{nullptr, nullptr, nullptr, nullptr, (T)0, nullptr};
    chain_pv_body<T>(L, seg, nframes, R0, loc, init_pos, init_vel, out_pos, out_vel, none);
}

// World rows in ONE launch of two workgroups: workgroup 0 walks the rotation chain (chain_rot_kernel's body) and publishes the rows of
// R0 chunk by chunk; workgroup 1 turns them into the frames' terms (frame_kernel's arithmetic) and walks the p / v chain
// (finish_kernel's body) a chunk or two behind -- the two serial chains of a 5000-frame trajectory overlap (236 us + 165 us one
// after the other before).  A grid of two: both workgroups become resident (a busy XCD can only delay workgroup 0, and workgroup 1's wait is bounded).
template <class T>
__global__ __launch_bounds__(256) void chain_world_kernel(const int64_t* __restrict__ seg, int nframes, const T* __restrict__ ir,
                                                           const T* __restrict__ init_rot, T* __restrict__ R0, T* __restrict__ rot_copy,
                                                           const T* __restrict__ dt, const T* __restrict__ acc, T gravity, T* __restrict__ loc_out,
                                                           const T* __restrict__ init_pos, const T* __restrict__ init_vel,
                                                           T* __restrict__ out_pos, T* __restrict__ out_vel, int* __restrict__ ready) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    if (blockIdx.x == 0) {
        chain_rot_body<T>(*reinterpret_cast<RotLds<T>*>(dyn_lds), seg, nframes, ir, init_rot, R0, rot_copy, ready);
    } else {
        const FrameSrc<T> fr{ready, dt, acc, ir, gravity, loc_out};
        chain_pv_body<T>(*reinterpret_cast<ChainLds<T>*>(dyn_lds), seg, nframes, R0, nullptr, init_pos, init_vel, out_pos, out_vel, fr);
    }
}

template <class T>
int run(const T* dt, const T* gyro, const T* acc, const int64_t* seg, int nframes, int64_t S, const T* ip, const T* ir0,
        const T* iv, double gravity, int motion_mode, T* opos, T* orot, T* ovel, void* scratch, int maxF, hipStream_t s,
        T* mpos = nullptr, T* mrot = nullptr, T* mvel = nullptr) {
    T* ir = reinterpret_cast<T*>(scratch);                 // 4 * (S + nframes)
    T* R0 = ir + 4 * ((size_t)S + nframes);                // 4 * (nframes + 1)
    T* loc = R0 + 4 * ((size_t)nframes + 1);               // 7 * nframes
    const size_t lds = 2 * 4 * (size_t)(maxF + 1) * sizeof(T);
    if (lds > 64 * 1024) return fail(ISLAM_EARG, "islam_imu_preint: %d IMU samples in one frame interval exceed the LDS scan buffer", maxF);
    // world rows: the two serial chains in one launch of two workgroups (chain_world_kernel); ISLAM_IMU_FUSED_WORLD=0: one after the
    // other in three launches (A/B runs)
    static const bool fused_world = [] { const char* e = std::getenv("ISLAM_IMU_FUSED_WORLD"); return !(e && e[0] == '0'); }();
    int* ready = reinterpret_cast<int*>(align_up(reinterpret_cast<size_t>(loc + 7 * (size_t)nframes), 8));      // (inside the 256 spare bytes)
    const bool fw = fused_world && motion_mode != 1;
    hipLaunchKernelGGL(scan_kernel<T>, dim3(nframes), dim3(64), lds, s, dt, gyro, seg, ir, fw ? ready : (int*)nullptr);
    if (fw) {
        constexpr size_t dyn = sizeof(RotLds<T>) > sizeof(ChainLds<T>) ? sizeof(RotLds<T>) : sizeof(ChainLds<T>);
        static_assert(dyn <= 64 * 1024, "chain_world_kernel: LDS");
        hipLaunchKernelGGL(chain_world_kernel<T>, dim3(2), dim3(256), dyn, s, seg, nframes, ir, ir0, R0, orot, dt, acc, (T)gravity,
                           loc, ip, iv, opos, ovel, ready);      // (loc: the motion rows and the backward pass read it from the scratch)
    } else {
        hipLaunchKernelGGL(chain_rot_kernel<T>, dim3(1), dim3(256), 0, s, seg, nframes, ir, ir0, R0, motion_mode != 1 ? orot : (T*)nullptr);
        hipLaunchKernelGGL(frame_kernel<T>, dim3((nframes + 63) / 64), dim3(64), 0, s, dt, acc, seg, nframes, ir, R0, (T)gravity, loc);
    }
    // motion_mode 2: both sets of outputs from ONE scan / rotation chain / frame pass (opos / orot / ovel: world rows, then the
    // motion rows in mpos / mrot / mvel) -- the two modes differ in the last kernel only
    if (motion_mode != 0)
        hipLaunchKernelGGL(finish_kernel<T>, dim3((nframes + 63) / 64), dim3(64), 0, s, seg, nframes, R0, loc, ip, iv, 1,
                           motion_mode == 2 ? mpos : opos, motion_mode == 2 ? mrot : orot, motion_mode == 2 ? mvel : ovel);
    if (motion_mode != 1 && !fw)
        hipLaunchKernelGGL(finish_kernel<T>, dim3(1), dim3(256), 0, s, seg, nframes, R0, loc, ip, iv, 0, opos, orot, ovel);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

// ---------------------------------------------------------------------------------------------------- backward
// Reverse-mode derivative of the whole frame loop w.r.t. the gyro and accelerometer samples (what PyPose's autograd gives the
// reference through pp.module.IMUPreintegrator when the denoiser runs with grad enabled: SURVEY F6 / section 8f rank 4).
// Rotations are differentiated in PyPose's convention: the gradient of a quaternion output / input is a LEFT-perturbation
// tangent vector (Exp(d) * R) in its first three slots, fourth slot 0.  With A_j = incre_r[j],
// R_i = rotation at the start of frame i, w_j = gyro_j dt_j, Q_{j+1} = R_i A_{j+1}, a_j = acc_j - Q_{j+1}^T g, ra_j = A_j a_j:
//   V = sum ra_j d_j, P = sum_j (V_j d_j + ra_j d_j^2 / 2);   frame outputs R_i V, R_i P, A_F (motion) or the chained p/v/R.
//   d A_{j+1} = d A_j + A_j Jl(w_j) d w_j          d R_{i+1} = d R_i + R_i d A_F          (left perturbations)
// One workgroup: (A) one lane per frame walks its samples backwards carrying S = the adjoint of A_{j+1}; (B) one lane chains
// the adjoint of R_i backwards over the frames (world mode: of p and v too, before (A)); (C) one lane per frame adds the
// part of the gyro gradient that comes from later frames through R_{i+1}.  Arithmetic in double whatever the I/O type.
struct V3 { double x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 rot3(Q<double> q, V3 p) {
    double in[3] = {p.x, p.y, p.z}, o[3];
    qact<double>(q, in, o);
    return {o[0], o[1], o[2]};
}
__device__ __forceinline__ Q<double> conj(Q<double> q) { return {-q.x, -q.y, -q.z, q.w}; }
template <class T> __device__ __forceinline__ Q<double> ldqd(const T* p) { return {(double)p[0], (double)p[1], (double)p[2], (double)p[3]}; }
template <class T> __device__ __forceinline__ V3 ld3d(const T* p) { return p ? V3{(double)p[0], (double)p[1], (double)p[2]} : V3{0, 0, 0}; }
// Jl(w)^T u = Jl(-w) u
__device__ __forceinline__ V3 JlT(V3 w, V3 u) {
    const double th2 = w.x * w.x + w.y * w.y + w.z * w.z, th = sqrt(th2);
    double c1, c2;
    if (th > 1e-4) { c1 = (1.0 - cos(th)) / th2; c2 = (th - sin(th)) / (th2 * th); }
    else { c1 = 0.5 - th2 / 24.0; c2 = 1.0 / 6.0 - th2 / 120.0; }
    const V3 wu = cross3(w, u);
    return u + (-c1) * wu + c2 * cross3(w, wu);
}

template <class T>
__global__ __launch_bounds__(256) void preint_bwd_kernel(const T* __restrict__ dt, const T* __restrict__ gyro, const T* __restrict__ acc,
                                                          const int64_t* __restrict__ seg, int nframes, const T* __restrict__ ir,
                                                          const T* __restrict__ R0, const T* __restrict__ loc, double gravity,
                                                          int motion_mode, const T* __restrict__ g_pos, const T* __restrict__ g_rot,
                                                          const T* __restrict__ g_vel, T* __restrict__ g_gyro, T* __restrict__ g_acc,
                                                          double* __restrict__ ws) {
    // ws: gv (3n) | gp (3n) | gR local, then T_{i+1} (3n)
    double* gv = ws;
    double* gp = ws + 3 * (size_t)nframes;
    double* gR = gp + 3 * (size_t)nframes;
    const int tid = threadIdx.x;
    const V3 g{0, 0, gravity};
    // ---- upstream adjoints of the frame-local sums
    if (motion_mode) {
        for (int i = tid; i < nframes; i += 256) {
            const V3 a = ld3d(g_vel ? g_vel + 3 * (size_t)i : nullptr), b = ld3d(g_pos ? g_pos + 3 * (size_t)i : nullptr);
            gv[3 * i] = a.x; gv[3 * i + 1] = a.y; gv[3 * i + 2] = a.z;
            gp[3 * i] = b.x; gp[3 * i + 1] = b.y; gp[3 * i + 2] = b.z;
        }
    } else if (tid == 0) {                      // world mode: p_{i+1} = p_i + R_i P_i + v_i t_i ; v_{i+1} = v_i + R_i V_i (F_i > 0)
        V3 pb = ld3d(g_pos ? g_pos + 3 * (size_t)nframes : nullptr), vb = ld3d(g_vel ? g_vel + 3 * (size_t)nframes : nullptr);
        for (int i = nframes - 1; i >= 0; --i) {
            const int F = (int)(seg[i + 1] - seg[i]);
            gv[3 * i] = vb.x; gv[3 * i + 1] = vb.y; gv[3 * i + 2] = vb.z;
            gp[3 * i] = pb.x; gp[3 * i + 1] = pb.y; gp[3 * i + 2] = pb.z;
            const V3 gpi = ld3d(g_pos ? g_pos + 3 * (size_t)i : nullptr), gvi = ld3d(g_vel ? g_vel + 3 * (size_t)i : nullptr);
            if (F == 0) { vb = gvi; pb = gpi + pb; }            // velocity zeroed, position held (imu_integrator.py:134-140)
            else { vb = gvi + vb + (double)loc[7 * (size_t)i + 6] * pb; pb = gpi + pb; }
        }
    }
    __syncthreads();
    // ---- (A) per frame, samples backwards
    for (int i = tid; i < nframes; i += 256) {
        const int a0 = (int)seg[i], F = (int)(seg[i + 1] - seg[i]);
        V3 gRl{0, 0, 0};
        if (F > 0) {
            const Q<double> Ri = ldqd(R0 + 4 * (size_t)i), RiT = conj(Ri);
            const T* irf = ir + 4 * ((size_t)a0 + i);
            const V3 gvi{gv[3 * i], gv[3 * i + 1], gv[3 * i + 2]}, gpi{gp[3 * i], gp[3 * i + 1], gp[3 * i + 2]};
            const V3 Vb = rot3(RiT, gvi), Pb = rot3(RiT, gpi);
            const V3 RV = ld3d(loc + 7 * (size_t)i), RP = ld3d(loc + 7 * (size_t)i + 3);
            gRl = cross3(RV, gvi) + cross3(RP, gpi);
            V3 S = motion_mode ? ld3d(g_rot ? g_rot + 4 * (size_t)i : nullptr) : V3{0, 0, 0};      // adjoint of A_F
            V3 carry{0, 0, 0};                   // ra_{k+1} x rabar_{k+1}: belongs to the adjoint of A_{k+1}
            double Tk = 0.0;                     // sum of d_j, j > k
            for (int k = F - 1; k >= 0; --k) {
                const double d = (double)dt[a0 + k];
                const Q<double> Ak = ldqd(irf + 4 * k), Ak1 = ldqd(irf + 4 * (k + 1));
                const Q<double> Qk1 = qmul<double>(Ri, Ak1);
                const V3 gb = rot3(conj(Qk1), g);
                const V3 av{(double)acc[3 * (a0 + k)] - gb.x, (double)acc[3 * (a0 + k) + 1] - gb.y, (double)acc[3 * (a0 + k) + 2] - gb.z};
                const V3 ra = rot3(Ak, av);
                const V3 rab = d * Vb + (d * (0.5 * d + Tk)) * Pb;
                const V3 ab = rot3(conj(Ak), rab);
                g_acc[3 * (a0 + k)] = (T)ab.x; g_acc[3 * (a0 + k) + 1] = (T)ab.y; g_acc[3 * (a0 + k) + 2] = (T)ab.z;
                const V3 Qb = cross3(rot3(Qk1, (-1.0) * ab), g);          // adjoint of Q_{k+1} from gb = Q^T g, gbbar = -abar
                gRl = gRl + Qb;
                S = S + rot3(RiT, Qb) + carry;
                const V3 w{(double)gyro[3 * (a0 + k)] * d, (double)gyro[3 * (a0 + k) + 1] * d, (double)gyro[3 * (a0 + k) + 2] * d};
                const V3 wb = JlT(w, rot3(conj(Ak), S));
                g_gyro[3 * (a0 + k)] = (T)(wb.x * d); g_gyro[3 * (a0 + k) + 1] = (T)(wb.y * d); g_gyro[3 * (a0 + k) + 2] = (T)(wb.z * d);
                carry = cross3(ra, rab);
                Tk += d;
            }
        }
        gR[3 * i] = gRl.x; gR[3 * i + 1] = gRl.y; gR[3 * i + 2] = gRl.z;
    }
    __syncthreads();
    // ---- (B) adjoint of R_i chained backwards: gR[i] <- T_{i+1} (what frame i's A_F receives through R_{i+1})
    if (tid == 0) {
        V3 Tn = motion_mode ? V3{0, 0, 0} : ld3d(g_rot ? g_rot + 4 * (size_t)nframes : nullptr);
        for (int i = nframes - 1; i >= 0; --i) {
            const V3 local{gR[3 * i], gR[3 * i + 1], gR[3 * i + 2]};
            gR[3 * i] = Tn.x; gR[3 * i + 1] = Tn.y; gR[3 * i + 2] = Tn.z;
            Tn = local + Tn + (motion_mode ? V3{0, 0, 0} : ld3d(g_rot ? g_rot + 4 * (size_t)i : nullptr));
        }
    }
    __syncthreads();
    // ---- (C) the share of the gyro gradient that arrives through R_{i+1} = R_i A_F
    for (int i = tid; i < nframes; i += 256) {
        const int a0 = (int)seg[i], F = (int)(seg[i + 1] - seg[i]);
        const V3 Tn{gR[3 * i], gR[3 * i + 1], gR[3 * i + 2]};
        if (F == 0 || (Tn.x == 0.0 && Tn.y == 0.0 && Tn.z == 0.0)) continue;
        const V3 extra = rot3(conj(ldqd(R0 + 4 * (size_t)i)), Tn);
        const T* irf = ir + 4 * ((size_t)a0 + i);
        for (int k = 0; k < F; ++k) {
            const double d = (double)dt[a0 + k];
            const V3 w{(double)gyro[3 * (a0 + k)] * d, (double)gyro[3 * (a0 + k) + 1] * d, (double)gyro[3 * (a0 + k) + 2] * d};
            const V3 wb = JlT(w, rot3(conj(ldqd(irf + 4 * k)), extra));
            g_gyro[3 * (a0 + k)] += (T)(wb.x * d); g_gyro[3 * (a0 + k) + 1] += (T)(wb.y * d); g_gyro[3 * (a0 + k) + 2] += (T)(wb.z * d);
        }
    }
}

template <class T>
int run_bwd(const T* dt, const T* gyro, const T* acc, const int64_t* seg, int nframes, int64_t S, double gravity, int motion_mode,
            const void* fwd_scratch, const T* g_pos, const T* g_rot, const T* g_vel, T* g_gyro, T* g_acc, void* scratch, hipStream_t s) {
    const T* ir = reinterpret_cast<const T*>(fwd_scratch);
    const T* R0 = ir + 4 * ((size_t)S + nframes);
    const T* loc = R0 + 4 * ((size_t)nframes + 1);
    hipLaunchKernelGGL(preint_bwd_kernel<T>, dim3(1), dim3(256), 0, s, dt, gyro, acc, seg, nframes, ir, R0, loc, gravity, motion_mode,
                       g_pos, g_rot, g_vel, g_gyro, g_acc, reinterpret_cast<double*>(scratch));
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

}  // namespace

extern "C" {

size_t islam_imu_preint_bwd_scratch_bytes(int nframes) { return sizeof(double) * 9 * (size_t)(nframes > 0 ? nframes : 0) + 256; }

int islam_imu_preint_bwd(const void* dt, const void* gyro, const void* acc, const int64_t* seg, int nframes, int64_t S, double gravity,
                         int motion_mode, const void* fwd_scratch, const void* g_pos, const void* g_rot, const void* g_vel,
                         void* g_gyro, void* g_acc, void* scratch, int dtype, void* stream) {
    if (nframes < 1 || S < 0 || !fwd_scratch || !g_gyro || !g_acc || !scratch)
        return fail(ISLAM_EARG, "islam_imu_preint_bwd: bad argument (nframes=%d S=%lld)", nframes, (long long)S);
    hipStream_t s = as_stream(stream);
    const int flag = motion_mode ? 1 : 0;
    if (dtype == ISLAM_F64)
        return run_bwd<double>((const double*)dt, (const double*)gyro, (const double*)acc, seg, nframes, S, gravity, flag, fwd_scratch,
                               (const double*)g_pos, (const double*)g_rot, (const double*)g_vel, (double*)g_gyro, (double*)g_acc, scratch, s);
    if (dtype == ISLAM_F32)
        return run_bwd<float>((const float*)dt, (const float*)gyro, (const float*)acc, seg, nframes, S, gravity, flag, fwd_scratch,
                              (const float*)g_pos, (const float*)g_rot, (const float*)g_vel, (float*)g_gyro, (float*)g_acc, scratch, s);
    return fail(ISLAM_EARG, "islam_imu_preint_bwd: dtype %d", dtype);
}

size_t islam_imu_scratch_bytes(int64_t S, int nframes, int dtype) {
    const size_t es = dtype == ISLAM_F64 ? 8 : 4;
    return es * (4 * ((size_t)S + nframes) + 4 * ((size_t)nframes + 1) + 7 * (size_t)nframes) + 256;
}

// max_frame_samples: the largest seg[i+1]-seg[i]; the caller knows it (host-side rgb2imu_sync), passing it
// avoids a device->host read.
int islam_imu_preint(const void* dt, const void* gyro, const void* acc, const int64_t* seg, int nframes, int64_t S,
                     int max_frame_samples, const void* init_pos, const void* init_rot, const void* init_vel,
                     double gravity, int motion_mode, void* out_pos, void* out_rot, void* out_vel, void* scratch,
                     int dtype, void* stream) {
    if (nframes < 1 || S < 0) return fail(ISLAM_EARG, "islam_imu_preint: nframes=%d S=%lld", nframes, (long long)S);
    const int flag = motion_mode ? 1 : 0, maxF = max_frame_samples;
    if (maxF < 0 || maxF > S) return fail(ISLAM_EARG, "islam_imu_preint: max frame samples %d out of range", maxF);
    hipStream_t s = as_stream(stream);
    if (dtype == ISLAM_F64)
        return run<double>((const double*)dt, (const double*)gyro, (const double*)acc, seg, nframes, S, (const double*)init_pos,
                           (const double*)init_rot, (const double*)init_vel, gravity, flag, (double*)out_pos, (double*)out_rot,
                           (double*)out_vel, scratch, maxF, s);
    if (dtype == ISLAM_F32)
        return run<float>((const float*)dt, (const float*)gyro, (const float*)acc, seg, nframes, S, (const float*)init_pos,
                          (const float*)init_rot, (const float*)init_vel, gravity, flag, (float*)out_pos, (float*)out_rot,
                          (float*)out_vel, scratch, maxF, s);
    return fail(ISLAM_EARG, "islam_imu_preint: dtype %d", dtype);
}

// Both call forms of IMUModule.integrate on the same frame range (the reference's loop calls it twice per batch, train.py:200-215:
// world rows for the trajectory, motion rows for the PVGO factors) from ONE pass: the sample scan, the rotation chain and the
// frame sums are the same in both modes (imu_integrator.py:116-158 with the same init['rot']), only the last step differs.
// world_*: nframes + 1 rows (row 0 = the initial state); motion_*: nframes rows (p0 = v0 = 0).  Bit-identical to the two
// islam_imu_preint calls.
int islam_imu_preint_both(const void* dt, const void* gyro, const void* acc, const int64_t* seg, int nframes, int64_t S,
                          int max_frame_samples, const void* init_pos, const void* init_rot, const void* init_vel, double gravity,
                          void* world_pos, void* world_rot, void* world_vel, void* motion_pos, void* motion_rot, void* motion_vel,
                          void* scratch, int dtype, void* stream) {
    if (nframes < 1 || S < 0) return fail(ISLAM_EARG, "islam_imu_preint_both: nframes=%d S=%lld", nframes, (long long)S);
    const int maxF = max_frame_samples;
    if (maxF < 0 || maxF > S) return fail(ISLAM_EARG, "islam_imu_preint_both: max frame samples %d out of range", maxF);
    hipStream_t s = as_stream(stream);
    if (dtype == ISLAM_F64)
        return run<double>((const double*)dt, (const double*)gyro, (const double*)acc, seg, nframes, S, (const double*)init_pos,
                           (const double*)init_rot, (const double*)init_vel, gravity, 2, (double*)world_pos, (double*)world_rot,
                           (double*)world_vel, scratch, maxF, s, (double*)motion_pos, (double*)motion_rot, (double*)motion_vel);
    if (dtype == ISLAM_F32)
        return run<float>((const float*)dt, (const float*)gyro, (const float*)acc, seg, nframes, S, (const float*)init_pos,
                          (const float*)init_rot, (const float*)init_vel, gravity, 2, (float*)world_pos, (float*)world_rot,
                          (float*)world_vel, scratch, maxF, s, (float*)motion_pos, (float*)motion_rot, (float*)motion_vel);
    return fail(ISLAM_EARG, "islam_imu_preint_both: dtype %d", dtype);
}

}  // extern "C"
