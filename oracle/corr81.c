/* Oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C restatement of the reference's 81-channel local correlation and its two gradients
 *   forward      : Network/PWC/correlation.py:35-103 (kernel_Correlation_updateOutput) after the
 *                  zero-padded rearrange :8-33
 *   grad first   : Network/PWC/correlation.py:105-167
 *   grad second  : Network/PWC/correlation.py:169-233
 * and of PWCDCNet.warp (Network/PWC/PWCNet.py:170-206): bilinear grid_sample(align_corners=True,
 * zeros padding) times the validity mask (grid_sample(ones) >= 0.9999).
 * Inputs are contiguous NCHW float32; accumulation in double (the HIP kernels accumulate in
 * float32 in a different order, so tests compare with a tolerance).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

void islam_oracle_corr81_fwd(const float* f1, const float* f2, float* out, int B, int C, int H, int W) {
    for (int b = 0; b < B; ++b)
        for (int dy = -4; dy <= 4; ++dy)
            for (int dx = -4; dx <= 4; ++dx) {
                int ch = (dy + 4) * 9 + (dx + 4);
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < W; ++x) {
                        int y2 = y + dy, x2 = x + dx;
                        double s = 0.0;
                        if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W)
                            for (int c = 0; c < C; ++c)
                                s += (double)f1[((size_t)(b * C + c) * H + y) * W + x] *
                                     (double)f2[((size_t)(b * C + c) * H + y2) * W + x2];
                        out[((size_t)(b * 81 + ch) * H + y) * W + x] = (float)(s / (double)C);
                    }
            }
}

/* gF[b,c,y,x] = (1/C) sum_{p,o} gOut[b,(p+4)*9+(o+4),y,x] * f2pad[b,c,y+p,x+o] */
void islam_oracle_corr81_bwd_first(const float* f2, const float* gout, float* g1, int B, int C, int H, int W) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    double s = 0.0;
                    for (int p = -4; p <= 4; ++p)
                        for (int o = -4; o <= 4; ++o) {
                            int y2 = y + p, x2 = x + o;
                            if (y2 < 0 || y2 >= H || x2 < 0 || x2 >= W) continue;
                            s += (double)gout[((size_t)(b * 81 + (p + 4) * 9 + (o + 4)) * H + y) * W + x] *
                                 (double)f2[((size_t)(b * C + c) * H + y2) * W + x2];
                        }
                    g1[((size_t)(b * C + c) * H + y) * W + x] = (float)(s / (double)C);
                }
}

/* gS[b,c,y,x] = (1/C) sum_{p,o} gOut[b,op,y-p,x-o] * f1[b,c,y-p,x-o]   (bounds-checked) */
void islam_oracle_corr81_bwd_second(const float* f1, const float* gout, float* g2, int B, int C, int H, int W) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    double s = 0.0;
                    for (int p = -4; p <= 4; ++p)
                        for (int o = -4; o <= 4; ++o) {
                            int y1 = y - p, x1 = x - o;
                            if (y1 < 0 || y1 >= H || x1 < 0 || x1 >= W) continue;
                            s += (double)gout[((size_t)(b * 81 + (p + 4) * 9 + (o + 4)) * H + y1) * W + x1] *
                                 (double)f1[((size_t)(b * C + c) * H + y1) * W + x1];
                        }
                    g2[((size_t)(b * C + c) * H + y) * W + x] = (float)(s / (double)C);
                }
}

/* PWCDCNet.warp: out = grid_sample(x, grid+flow) * (grid_sample(1, grid+flow) >= 0.9999).
 * The normalise/un-normalise round trip of grid_sample(align_corners=True) is restated in float32:
 *   gx = 2*(x+fx)/max(W-1,1) - 1 ;  ix = ((gx+1)/2)*(W-1). */
void islam_oracle_warp(const float* x, const float* flo, float* out, int B, int C, int H, int W) {
    for (int b = 0; b < B; ++b)
        for (int yy = 0; yy < H; ++yy)
            for (int xx = 0; xx < W; ++xx) {
                float vx = (float)xx + flo[((size_t)(b * 2 + 0) * H + yy) * W + xx];
                float vy = (float)yy + flo[((size_t)(b * 2 + 1) * H + yy) * W + xx];
                float gx = 2.0f * vx / (float)(W > 1 ? W - 1 : 1) - 1.0f;
                float gy = 2.0f * vy / (float)(H > 1 ? H - 1 : 1) - 1.0f;
                float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
                float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
                float x0f = floorf(ix), y0f = floorf(iy);
                int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
                float x1f = x0f + 1.0f, y1f = y0f + 1.0f;      /* torch GridSampler: nw=(x_se-ix)(y_se-iy) ... */
                float w00 = (x1f - ix) * (y1f - iy), w01 = (ix - x0f) * (y1f - iy);
                float w10 = (x1f - ix) * (iy - y0f), w11 = (ix - x0f) * (iy - y0f);
                int v00 = (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H), v01 = (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H);
                int v10 = (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H), v11 = (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H);
                float m = 0.0f;
                if (v00) m += w00;
                if (v01) m += w01;
                if (v10) m += w10;
                if (v11) m += w11;
                float mask = (m < 0.9999f) ? 0.0f : 1.0f;
                for (int c = 0; c < C; ++c) {
                    const float* p = x + (size_t)(b * C + c) * H * W;
                    float s = 0.0f;
                    if (v00) s += p[y0 * W + x0] * w00;
                    if (v01) s += p[y0 * W + x1] * w01;
                    if (v10) s += p[y1 * W + x0] * w10;
                    if (v11) s += p[y1 * W + x1] * w11;
                    out[((size_t)(b * C + c) * H + yy) * W + xx] = s * mask;
                }
            }
}
