// Stereo scale recovery on gfx950: one fused pass per image.
//
// Replaces the per-sample Python loop TartanVO.py:159-167 around dense_ba.py:88-176
// (scale_from_disp_flow, disparity branch: ~40 tiny kernels and a boolean-index compaction with a
// device sync per sample) and is_inside_image* (dense_ba.py:65-72, inclusive upper bound).
//
// Per pixel: masks, depth z = fx*b/disp, back-projection P = z K^-1 [u v 1], rows of the 1-DoF system
//   M1 = a2 fu - a0, w1 = b0 - b2 fu, M2 = a2 fv - a1, w2 = b1 - b2 fv,  a = K t^, b = K R P, f = flow + uv
// and the reductions  s = (M^T w) / (M^T M).  HBM-bound: reads disp, flow (3 floats) + edge byte,
// writes z + two mask bytes per pixel; the sums needed for d s / d pose are accumulated in the same
// pass so the backward pass never re-reads the images:
//   sums[0] = sum M^2            sums[1] = sum M w
//   sums[2] = sum w1             sums[3] = sum w2            sums[4] = sum (w1 fu + w2 fv)
//   sums[5] = sum M1             sums[6] = sum M2            sums[7] = sum (M1 fu + M2 fv)
//   sums[8..10]  = sum M1 P      sums[11..13] = sum M2 P     sums[14..16] = sum (M1 (cx-fu) + M2 (cy-fv)) P
//   sums[17] = number of pixels in the mask
#include <hip/hip_runtime.h>

#include "common.h"

using namespace islam;

namespace {

constexpr int NS = ISLAM_SCALE_NSUM;
constexpr int NBLK = ISLAM_SCALE_NBLK;

__global__ __launch_bounds__(256) void scale_partial_kernel(const float* __restrict__ disp, const float* __restrict__ flow,
                                                             const float* __restrict__ pose7, const float* __restrict__ intr4,
                                                             const float* __restrict__ baseline, const uint8_t* __restrict__ edge,
                                                             const float* __restrict__ disp_th, float* __restrict__ z,
                                                             uint8_t* __restrict__ mask, uint8_t* __restrict__ dmask,
                                                             double* __restrict__ partial, int H, int W, int depth_input) {
    const int b = blockIdx.y;
    const int HW = H * W;
    const float fx = intr4[4 * b], fy = intr4[4 * b + 1], cx = intr4[4 * b + 2], cy = intr4[4 * b + 3];
    const float bl = baseline[b], dth = disp_th[b];
    // T^-1 of the motion: R = R(q)^T, t = -R q.t ; t^ = t/|t|
    const float* ps = pose7 + 7 * b;
    const float qx = -ps[3], qy = -ps[4], qz = -ps[5], qw = ps[6];
    const float r00 = 1 - 2 * (qy * qy + qz * qz), r01 = 2 * (qx * qy - qz * qw), r02 = 2 * (qx * qz + qy * qw);
    const float r10 = 2 * (qx * qy + qz * qw), r11 = 1 - 2 * (qx * qx + qz * qz), r12 = 2 * (qy * qz - qx * qw);
    const float r20 = 2 * (qx * qz - qy * qw), r21 = 2 * (qy * qz + qx * qw), r22 = 1 - 2 * (qx * qx + qy * qy);
    const float t0 = -(r00 * ps[0] + r01 * ps[1] + r02 * ps[2]);
    const float t1 = -(r10 * ps[0] + r11 * ps[1] + r12 * ps[2]);
    const float t2 = -(r20 * ps[0] + r21 * ps[1] + r22 * ps[2]);
    const float tn = fmaxf(sqrtf(t0 * t0 + t1 * t1 + t2 * t2), 1e-12f);
    const float n0 = t0 / tn, n1 = t1 / tn, n2 = t2 / tn;
    const float a0 = fx * n0 + cx * n2, a1 = fy * n1 + cy * n2, a2 = n2;

    double acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] = 0.0;

    const float* dp = disp + (size_t)b * HW;
    const float* fxp = flow + (size_t)b * 2 * HW;
    const float* fyp = fxp + HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += NBLK * 256) {
        const int v = i / W, u = i - v * W;
        const float uf = (float)u, vf = (float)v;
        const float d = dp[i], flx = fxp[i], fly = fyp[i];
        const float fu = flx + uf, fv = fly + vf;
        const bool inside = fu >= 0.f && fu <= (float)W && fv >= 0.f && fv <= (float)H;     // inclusive upper bound (Q6)
        bool m = inside && (sqrtf(flx * flx + fly * fly) > 0.f);
        if (edge) m = m && (edge[(size_t)b * HW + i] != 0);
        bool dm;
        float zz;
        if (depth_input) {                 // dense_ba.py:125-131: `disp` holds depth, valid up to fx*baseline (disparity >= 1)
            dm = d <= fx * bl && d > 0.f;
            zz = dm ? d : 0.f;
        } else {                           // dense_ba.py:115-123
            const float du = -d + uf;
            dm = du >= 0.f && du <= (float)W && d >= dth;
            zz = dm ? fx * bl / d : 0.f;
        }
        m = m && dm;
        z[(size_t)b * HW + i] = zz;
        mask[(size_t)b * HW + i] = m ? 1 : 0;
        dmask[(size_t)b * HW + i] = dm ? 1 : 0;
        if (m) {
            const float Px = zz * ((uf - cx) / fx), Py = zz * ((vf - cy) / fy), Pz = zz;
            const float q0 = r00 * Px + r01 * Py + r02 * Pz;
            const float q1 = r10 * Px + r11 * Py + r12 * Pz;
            const float q2 = r20 * Px + r21 * Py + r22 * Pz;
            const float b0 = fx * q0 + cx * q2, b1 = fy * q1 + cy * q2, b2 = q2;
            const double M1 = a2 * fu - a0, w1 = b0 - b2 * fu, M2 = a2 * fv - a1, w2 = b1 - b2 * fv;
            const double k3 = M1 * (double)(cx - fu) + M2 * (double)(cy - fv);
            acc[0] += M1 * M1 + M2 * M2;
            acc[1] += M1 * w1 + M2 * w2;
            acc[2] += w1; acc[3] += w2; acc[4] += w1 * fu + w2 * fv;
            acc[5] += M1; acc[6] += M2; acc[7] += M1 * fu + M2 * fv;
            acc[8] += M1 * Px; acc[9] += M1 * Py; acc[10] += M1 * Pz;
            acc[11] += M2 * Px; acc[12] += M2 * Py; acc[13] += M2 * Pz;
            acc[14] += k3 * Px; acc[15] += k3 * Py; acc[16] += k3 * Pz;
            acc[17] += 1.0;
        }
    }
    __shared__ double red[4][NS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wv][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS)
        partial[((size_t)b * NBLK + blockIdx.x) * NS + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void scale_final_kernel(const double* __restrict__ partial, double* __restrict__ sums, float* __restrict__ scale, int B) {
    const int b = blockIdx.x, i = threadIdx.x;
    __shared__ double s[NS];
    if (i < NS) {
        double v = 0.0;
        for (int k = 0; k < NBLK; ++k) v += partial[((size_t)b * NBLK + k) * NS + i];
        sums[(size_t)b * NS + i] = v;
        s[i] = v;
    }
    __syncthreads();
    if (i == 0) scale[b] = (float)(1.0 / s[0] * s[1]);     // s = 1/sum(M*M) * M^T w ; empty mask -> NaN like the reference (Q7)
}

}  // namespace

static int scale_ls_launch(const float* disp, const float* flow, const float* pose7, const float* intr4, const float* baseline,
                           const uint8_t* edge, const float* disp_th, float* scale, float* z, uint8_t* mask, uint8_t* dmask,
                           double* sums, double* partial, int B, int H, int W, int depth_input, void* stream) {
    if (B < 1 || H < 1 || W < 1) return fail(ISLAM_EARG, "islam_scale_ls: bad shape (%d,%d,%d)", B, H, W);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(scale_partial_kernel, dim3(NBLK, B), dim3(256), 0, s, disp, flow, pose7, intr4, baseline, edge, disp_th,
                       z, mask, dmask, partial, H, W, depth_input);
    hipLaunchKernelGGL(scale_final_kernel, dim3(B), dim3(64), 0, s, partial, sums, scale, B);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}

extern "C" int islam_scale_ls(const float* disp, const float* flow, const float* pose7, const float* intr4,
                              const float* baseline, const uint8_t* edge, const float* disp_th, float* scale, float* z,
                              uint8_t* mask, uint8_t* dmask, double* sums, double* partial, int B, int H, int W, void* stream) {
    return scale_ls_launch(disp, flow, pose7, intr4, baseline, edge, disp_th, scale, z, mask, dmask, sums, partial, B, H, W, 0,
                           stream);
}

// the depth-input branch of scale_from_disp_flow (dense_ba.py:125-131): `depth` replaces the disparity map
extern "C" int islam_scale_ls_depth(const float* depth, const float* flow, const float* pose7, const float* intr4,
                                    const float* baseline, const uint8_t* edge, float* scale, float* z, uint8_t* mask,
                                    uint8_t* dmask, double* sums, double* partial, int B, int H, int W, void* stream) {
    return scale_ls_launch(depth, flow, pose7, intr4, baseline, edge, baseline /* unused */, scale, z, mask, dmask, sums, partial,
                           B, H, W, 1, stream);
}
