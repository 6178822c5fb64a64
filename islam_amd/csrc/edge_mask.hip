// Edge mask of TartanVO.forward on gfx950: ONE launch, one workgroup per image, the whole quarter-resolution image in LDS.
//
// Replaces reference TartanVO.py:145-155 -- a 27.5 MB device->host copy of img0 (B=8), then per image on the host
// cv2.resize(1/4) -> cv2.Canny(50, 100) -> cv2.dilate(5x5 ones) -> `> 0`, then a host->device copy -- which forces a full
// device synchronisation in the middle of the forward.  Here nothing leaves the device and the host never waits.
//
// Integer arithmetic of OpenCV 4.7 (the version the reference pins, environment.yml:149), restated in oracle/canny.py:
//   u8      : (uint8)(img * 255)                       numpy astype truncation
//   resize  : INTER_LINEAR at exactly 1/4 = (a + b + c + d + 2) >> 2 over the centre 2x2 of every 4x4 cell
//   Sobel   : 3x3, replicated border, per channel; the first channel with the largest |dx| + |dy| wins
//   NMS     : only where mag > low; sector test with TG22 = 13573 (tan 22.5 deg in Q15); asymmetric > / >= comparisons
//   double threshold + 8-connected hysteresis (the stack-based flood fill of OpenCV reaches the same fixed point as the
//   in-LDS relaxation below: edges = the candidates connected to a strong pixel)
//   dilate  : 5x5 ones, border never wins
//
// Workgroup = 1024 threads; n = h*w <= 27 000 quarter-resolution pixels (112x160 = 17 920 for the 448x640 input):
//   LDS: img[3][n] u8 (later: map[n], tmp[n]) | mag[n] u16 (bits 0..11 magnitude <= 2040, bits 12..13 sector)
// Hysteresis: every thread owns a run of consecutive pixels in row-major order AND a run in column-major order and sweeps
// each forwards and backwards per round (so straight or diagonal chains of weak pixels are followed in one round, not one
// pixel per round); rounds repeat until a round promotes nothing.  Promotions are monotone (0 -> 2 only), so concurrent
// sweeps can only speed convergence up, never change the fixed point.
#include <hip/hip_runtime.h>

#include "common.h"

using namespace islam;

namespace {

constexpr int EDGE_THREADS = 1024;
constexpr int EDGE_MAX_PIXELS = 27000;          // 6 bytes of LDS per pixel + slack below the 160 KB of a CU
constexpr int TG22 = 13573;

__device__ __forceinline__ int to_u8(float v) { return (int)(uint8_t)(int)(v * 255.0f); }

__global__ __launch_bounds__(EDGE_THREADS) void edge_mask_kernel(const float* __restrict__ img, uint8_t* __restrict__ out,
                                                                  int H, int W, int downscale, int low, int high) {
    extern __shared__ __align__(16) uint8_t lds[];
    const int h = downscale ? H / 4 : H, w = downscale ? W / 4 : W, n = h * w;
    uint8_t* im = lds;                          // [3][n]
    uint16_t* mag = reinterpret_cast<uint16_t*>(lds + (size_t)((3 * n + 15) & ~15));
    uint8_t* map = im;                          // aliases channel 0 once the gradients are formed
    uint8_t* tmp = im + n;                      // aliases channel 1
    __shared__ int changed;
    const int tid = threadIdx.x;
    const float* src = img + (size_t)blockIdx.x * 3 * H * W;

    // ---- 1. uint8 conversion (+ quarter resize)
    for (int p = tid; p < n; p += EDGE_THREADS) {
        const int y = p / w, x = p - y * w;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* s = src + (size_t)c * H * W;
            int v;
            if (downscale) {
                const float* r0 = s + (size_t)(4 * y + 1) * W + 4 * x + 1;
                const float* r1 = r0 + W;
                v = (to_u8(r0[0]) + to_u8(r0[1]) + to_u8(r1[0]) + to_u8(r1[1]) + 2) >> 2;
            } else {
                v = to_u8(s[p]);
            }
            im[c * n + p] = (uint8_t)v;
        }
    }
    __syncthreads();

    // ---- 2. Sobel per channel, channel selection, sector code.  Results stay in registers until every thread has read
    //         its neighbourhood (mag does not alias im, but map / tmp do).
    for (int p = tid; p < n; p += EDGE_THREADS) {
        const int y = p / w, x = p - y * w;
        const int ym = y > 0 ? y - 1 : 0, yp = y < h - 1 ? y + 1 : h - 1;
        const int xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
        int bm = -1, bdx = 0, bdy = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint8_t* q = im + c * n;
            const int tl = q[ym * w + xm], tc = q[ym * w + x], tr = q[ym * w + xp];
            const int ml = q[y * w + xm], mr = q[y * w + xp];
            const int bl = q[yp * w + xm], bc = q[yp * w + x], br = q[yp * w + xp];
            const int dx = (tr + 2 * mr + br) - (tl + 2 * ml + bl);
            const int dy = (bl + 2 * bc + br) - (tl + 2 * tc + tr);
            const int m = abs(dx) + abs(dy);
            if (m > bm) { bm = m; bdx = dx; bdy = dy; }          // strict >: the first channel with the maximum
        }
        const int ax = abs(bdx), ay = abs(bdy) << 15;
        const int tg22x = ax * TG22;
        int sector;
        if (ay < tg22x) sector = 0;
        else if (ay > tg22x + (ax << 16)) sector = 1;
        else sector = ((bdx ^ bdy) < 0) ? 3 : 2;
        mag[p] = (uint16_t)(bm | (sector << 12));
    }
    __syncthreads();

    // ---- 3. non-maximum suppression + double threshold -> map: 2 edge, 0 candidate, 1 neither
    for (int p = tid; p < n; p += EDGE_THREADS) {
        const int y = p / w, x = p - y * w;
        const int v = mag[p], m = v & 0xFFF, sector = v >> 12;
        auto M = [&](int yy, int xx) -> int { return (yy < 0 || yy >= h || xx < 0 || xx >= w) ? 0 : (mag[yy * w + xx] & 0xFFF); };
        uint8_t r = 1;
        if (m > low) {
            bool keep;
            if (sector == 0) keep = m > M(y, x - 1) && m >= M(y, x + 1);
            else if (sector == 1) keep = m > M(y - 1, x) && m >= M(y + 1, x);
            else {
                const int s = sector == 3 ? -1 : 1;
                keep = m > M(y - 1, x - s) && m > M(y + 1, x + s);
            }
            if (keep) r = m > high ? 2 : 0;
        }
        map[p] = r;
    }
    __syncthreads();

    // ---- 4. hysteresis
    const int chunk = (n + EDGE_THREADS - 1) / EDGE_THREADS;
    const int p0 = tid * chunk, p1 = min(n, p0 + chunk);
    auto strong_near = [&](int y, int x) -> bool {
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if ((dy | dx) != 0 && yy >= 0 && yy < h && xx >= 0 && xx < w && map[yy * w + xx] == 2) return true;
            }
        return false;
    };
    for (int round = 0; round < n; ++round) {
        if (tid == 0) changed = 0;
        __syncthreads();
        int mine = 0;
        for (int dir = 0; dir < 2; ++dir)                       // row-major run: forwards, then backwards
            for (int k = 0; k < p1 - p0; ++k) {
                const int p = dir == 0 ? p0 + k : p1 - 1 - k;
                if (map[p] == 0) {
                    const int y = p / w, x = p - y * w;
                    if (strong_near(y, x)) { map[p] = 2; mine = 1; }
                }
            }
        for (int dir = 0; dir < 2; ++dir)                       // column-major run: downwards, then upwards
            for (int k = 0; k < p1 - p0; ++k) {
                const int c = dir == 0 ? p0 + k : p1 - 1 - k;
                const int x = c / h, y = c - x * h;
                if (map[y * w + x] == 0 && strong_near(y, x)) { map[y * w + x] = 2; mine = 1; }
            }
        if (mine) changed = 1;
        __syncthreads();
        const int again = changed;
        __syncthreads();
        if (!again) break;
    }

    // ---- 5. 5x5 dilation, separable: rows into tmp, columns to the output
    for (int p = tid; p < n; p += EDGE_THREADS) {
        const int y = p / w, x = p - y * w;
        uint8_t r = 0;
#pragma unroll
        for (int d = -2; d <= 2; ++d) {
            const int xx = x + d;
            if (xx >= 0 && xx < w && map[y * w + xx] == 2) r = 1;
        }
        tmp[p] = r;
    }
    __syncthreads();
    uint8_t* dst = out + (size_t)blockIdx.x * n;
    for (int p = tid; p < n; p += EDGE_THREADS) {
        const int y = p / w, x = p - y * w;
        uint8_t r = 0;
#pragma unroll
        for (int d = -2; d <= 2; ++d) {
            const int yy = y + d;
            if (yy >= 0 && yy < h && tmp[yy * w + x]) r = 1;
        }
        dst[p] = r;
    }
}

}  // namespace

extern "C" int islam_edge_mask_max_pixels(void) { return EDGE_MAX_PIXELS; }

extern "C" int islam_edge_mask(const float* img, uint8_t* mask, int B, int H, int W, int downscale, int low, int high,
                               void* stream) {
    if (B < 0 || H <= 0 || W <= 0 || (B > 0 && (!img || !mask))) return fail(ISLAM_EARG, "islam_edge_mask: bad argument");
    if (downscale && (H % 4 || W % 4)) return fail(ISLAM_EARG, "islam_edge_mask: H=%d, W=%d are not multiples of 4", H, W);
    if (low > high) { const int t = low; low = high; high = t; }           // cv::Canny swaps the thresholds
    const int h = downscale ? H / 4 : H, w = downscale ? W / 4 : W;
    const long n = (long)h * w;
    if (n > EDGE_MAX_PIXELS)
        return fail(ISLAM_EARG, "islam_edge_mask: %dx%d = %ld pixels after the resize, at most %d fit the LDS of one CU", h, w, n,
                    EDGE_MAX_PIXELS);
    if (B == 0) return ISLAM_OK;
    const size_t lds = (size_t)((3 * n + 15) & ~15L) + 2 * (size_t)n;
    int dev = 0;
    ISLAM_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set[64] = {};                                         // per device: the attribute lives in the device's module
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        ISLAM_HIP_CHECK(hipFuncSetAttribute((const void*)edge_mask_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(edge_mask_kernel, dim3(B), dim3(EDGE_THREADS), lds, as_stream(stream), img, mask, H, W, downscale, low, high);
    ISLAM_LAUNCH_CHECK();
    return ISLAM_OK;
}
