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
                                                   const int64_t* __restrict__ seg, T* __restrict__ ir) {
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
constexpr int CHAIN_CHUNK = 512;
constexpr int ROT_CHUNK = 256;        // frames per staged chunk of the rotation chain (16 factors per frame)

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

template <class T>
__global__ __launch_bounds__(256) void chain_rot_kernel(const int64_t* __restrict__ seg, int nframes, const T* __restrict__ ir,
                                                         const T* __restrict__ init_rot, T* __restrict__ R0) {
    // sb[j][c][k]: factor of term k of output component c (x, y, z, w) for frame j's increment b = incre_r_j[F_j]:
    //   x: +b.x +b.w +b.z -b.y    y: +b.y -b.z +b.w +b.x    z: +b.z +b.y -b.x +b.w    w: +b.w -b.x -b.y -b.z      (qmul above)
    __shared__ __attribute__((aligned(16))) T sb[ROT_CHUNK][4][4];
    __shared__ T out[ROT_CHUNK + 1][4];  // R0[i+1]; row ROT_CHUNK: where the idle lanes of the walking wave store (no branch in the loop)
    __shared__ int has[ROT_CHUNK];       // F_i > 0
    const int tid = threadIdx.x;
    // lane c < 4 of wave 0 carries component c of the running rotation (x, y, z, w)
    T rc = tid < 4 ? init_rot[tid] : (T)0;
    if (tid < 4) R0[tid] = rc;
    for (int base = 0; base < nframes; base += ROT_CHUNK) {
        const int cnt = min(ROT_CHUNK, nframes - base);
        __syncthreads();
        for (int j = tid; j < cnt; j += 256) {
            const int i = base + j;
            const int a = (int)seg[i], F = (int)(seg[i + 1] - seg[i]);
            has[j] = F > 0;
            const T* src = ir + 4 * ((size_t)a + i + F);
            const T bx = src[0], by = src[1], bz = src[2], bw = src[3];
            T* t = &sb[j][0][0];
            t[0] = bx;  t[1] = bw;   t[2] = bz;   t[3] = -by;
            t[4] = by;  t[5] = -bz;  t[6] = bw;   t[7] = bx;
            t[8] = bz;  t[9] = by;   t[10] = -bx; t[11] = bw;
            t[12] = bw; t[13] = -bx; t[14] = -by; t[15] = -bz;
        }
        __syncthreads();
        if (tid < 64) {                    // (whole wave 0 executes the loop: DPP needs its quad's lanes active; lanes >= 4 carry zeros)
            const int c = tid & 3;
            T* op = tid < 4 ? &out[0][c] : &out[ROT_CHUNK][c];
            const int ostep = tid < 4 ? 4 : 0;
            T f0 = sb[0][c][0], f1 = sb[0][c][1], f2 = sb[0][c][2], f3 = sb[0][c][3];
            int h = has[0];
            for (int j = 0; j < cnt; ++j) {
                const int jn = j + 1 < cnt ? j + 1 : j;                 // next frame's factors: requested before this frame's arithmetic
                const T n0 = sb[jn][c][0], n1 = sb[jn][c][1], n2 = sb[jn][c][2], n3 = sb[jn][c][3];
                const int hn = has[jn];
                const T aw = quad_bcast<0xff>(rc), ax = quad_bcast<0x00>(rc), ay = quad_bcast<0x55>(rc), az = quad_bcast<0xaa>(rc);
                const T o = aw * f0 + ax * f1 + ay * f2 + az * f3;      // = qmul(r, b) component c, same operations in the same order
                rc = h ? o : rc;
                *op = rc;
                op += ostep;
                f0 = n0; f1 = n1; f2 = n2; f3 = n3; h = hn;
            }
        }
        __syncthreads();
        for (int j = tid; j < cnt * 4; j += 256) R0[4 * (size_t)(base + 1) + j] = (&out[0][0])[j];
    }
}

// C: per frame (one lane each): local integration, rotated into the frame's start orientation.
//    loc[i] = { R0_i incre_v[F] (3), R0_i incre_p[F] (3), incre_t (1) }
template <class T>
__global__ __launch_bounds__(64) void frame_kernel(const T* __restrict__ dt, const T* __restrict__ acc,
                                                    const int64_t* __restrict__ seg, int nframes,
                                                    const T* __restrict__ ir, const T* __restrict__ R0, T gravity,
                                                    T* __restrict__ loc) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nframes) return;
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
    T* o = loc + 7 * (size_t)i;
    o[0] = rv[0]; o[1] = rv[1]; o[2] = rv[2]; o[3] = rp[0]; o[4] = rp[1]; o[5] = rp[2]; o[6] = it;
}

// D: outputs.  world mode: sequential p/v chain (row 0 = init).  motion mode: every frame starts from p = v = 0.
template <class T>
__global__ void finish_kernel(const int64_t* __restrict__ seg, int nframes, const T* __restrict__ R0,
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
    // world mode: sequential p/v chain (three lanes, one per coordinate), per-frame terms staged through LDS by the whole workgroup
    __shared__ T sl[CHAIN_CHUNK][7];
    __shared__ T so[CHAIN_CHUNK][6];
    __shared__ int sF[CHAIN_CHUNK];
    __shared__ T carry[9];                 // p (3), v (3), held position sp (3)
    const int tid = threadIdx.x;
    if (blockIdx.x != 0) return;
    if (tid == 0) {
        for (int c = 0; c < 3; ++c) {
            carry[c] = init_pos[c]; carry[3 + c] = init_vel[c]; carry[6 + c] = init_pos[c];
            out_pos[c] = init_pos[c]; out_vel[c] = init_vel[c];
        }
        stq(ldq(R0), out_rot);
    }
    for (int base = 0; base < nframes; base += CHAIN_CHUNK) {
        const int cnt = min(CHAIN_CHUNK, nframes - base);
        __syncthreads();
        for (int j = tid; j < cnt; j += blockDim.x) sF[j] = (int)(seg[base + j + 1] - seg[base + j]);
        for (int j = tid; j < cnt * 7; j += blockDim.x) (&sl[0][0])[j] = loc[7 * (size_t)base + j];
        __syncthreads();
        if (tid < 3) {
            // the three coordinates are independent chains: lane c walks coordinate c (one lane doing all three spent 12 dependent
            // double operations + LDS round trips per frame: 0.46-0.9 ms for 5000 frames), the next frame's terms requested
            // before this frame's arithmetic
            const int c = tid;
            T p = carry[c], v = carry[3 + c], sp = carry[6 + c];
            T a = sl[0][c], b = sl[0][3 + c], t = sl[0][6];
            int F = sF[0];
            for (int j = 0; j < cnt; ++j) {
                const int jn = j + 1 < cnt ? j + 1 : j;
                const T an = sl[jn][c], bn = sl[jn][3 + c], tn = sl[jn][6];
                const int Fn = sF[jn];
                T sv;
                if (F == 0) {                               // imu_integrator.py:134-140: vel zeroed, pos / rot held
                    sv = 0;
                } else {
                    sv = v + a;
                    sp = p + b + v * t;
                }
                so[j][c] = sp; so[j][3 + c] = sv; p = sp; v = sv;
                a = an; b = bn; t = tn; F = Fn;
            }
            carry[c] = p; carry[3 + c] = v; carry[6 + c] = sp;
        }
        __syncthreads();
        for (int j = tid; j < cnt * 3; j += blockDim.x) {
            const int f = j / 3, c = j - 3 * f;
            out_pos[3 * (size_t)(base + 1) + j] = so[f][c];
            out_vel[3 * (size_t)(base + 1) + j] = so[f][3 + c];
        }
        for (int j = tid; j < cnt * 4; j += blockDim.x) out_rot[4 * (size_t)(base + 1) + j] = R0[4 * (size_t)(base + 1) + j];
    }
}

template <class T>
int run(const T* dt, const T* gyro, const T* acc, const int64_t* seg, int nframes, int64_t S, const T* ip, const T* ir0,
        const T* iv, double gravity, int motion_mode, T* opos, T* orot, T* ovel, void* scratch, int maxF, hipStream_t s) {
    T* ir = reinterpret_cast<T*>(scratch);                 // 4 * (S + nframes)
    T* R0 = ir + 4 * ((size_t)S + nframes);                // 4 * (nframes + 1)
    T* loc = R0 + 4 * ((size_t)nframes + 1);               // 7 * nframes
    const size_t lds = 2 * 4 * (size_t)(maxF + 1) * sizeof(T);
    if (lds > 64 * 1024) return fail(ISLAM_EARG, "islam_imu_preint: %d IMU samples in one frame interval exceed the LDS scan buffer", maxF);
    hipLaunchKernelGGL(scan_kernel<T>, dim3(nframes), dim3(64), lds, s, dt, gyro, seg, ir);
    hipLaunchKernelGGL(chain_rot_kernel<T>, dim3(1), dim3(256), 0, s, seg, nframes, ir, ir0, R0);
    hipLaunchKernelGGL(frame_kernel<T>, dim3((nframes + 63) / 64), dim3(64), 0, s, dt, acc, seg, nframes, ir, R0, (T)gravity, loc);
    if (motion_mode)
        hipLaunchKernelGGL(finish_kernel<T>, dim3((nframes + 63) / 64), dim3(64), 0, s, seg, nframes, R0, loc, ip, iv, 1, opos, orot, ovel);
    else
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

}  // extern "C"
