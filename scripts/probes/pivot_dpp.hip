// Pivot phase of one node step of the block-tridiagonal elimination (pvgo.hip, twisted_sweep): 9 pivots, 36 row updates on the 28
// columns a wavefront holds one per lane.  A: as shipped -- the multiplier of a row update comes from lane i through two
// v_readlane_b32.  B: DP-ALU DPP -- v_fmac_f64_dpp ... row_newbcast:i with the nine S columns replicated in every row of 16 lanes.
// Prints clocks per node step for one wave per SIMD and checks that both forms give the same bits.
//   hipcc --offload-arch=gfx950 -O3 -o pivot_dpp pivot_dpp.hip && ./pivot_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__device__ __forceinline__ double bcast(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rcp_nr(double p) {
    double r = __builtin_amdgcn_rcp(p);
    double e = fma(-p, r, 1.0);
    r = fma(r, e, r);
    e = fma(-p, r, 1.0);
    return fma(r, e, r);
}

template <int I> __device__ __forceinline__ double dpp_bcast(double v) {
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
// m += bcast_I(m) * nf
template <int I> __device__ __forceinline__ void dpp_update(double& m, double nf) {
    if constexpr (I == 0) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(nf));
    if constexpr (I == 1) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(nf));
    if constexpr (I == 2) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(nf));
    if constexpr (I == 3) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(nf));
    if constexpr (I == 4) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(nf));
    if constexpr (I == 5) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(nf));
    if constexpr (I == 6) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(nf));
    if constexpr (I == 7) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(nf));
}

template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

// lane -> column: A: lane (0..27); B: row k = lane / 16: lanes 16k + 0..8 = S column (replica k), 16k + 9..15 = other column 7k + (lane & 15) - 9
__device__ __forceinline__ int col_of(int lane, bool dpp) {
    if (!dpp) return lane < 28 ? lane : 27;
    const int k = lane >> 4, j = lane & 15;
    if (j < 9) return j;
    const int o = 7 * k + (j - 9);
    return o < 19 ? 9 + o : 27;
}

template <bool DPP>
__global__ __launch_bounds__(64) void pivots(const double* __restrict__ in, double* __restrict__ out, long long* __restrict__ clk, int reps) {
    const int lane = threadIdx.x, col = col_of(lane, DPP);
    double m0[9];
    for (int r = 0; r < 9; ++r) m0[r] = in[col * 9 + r];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < reps; ++it) {
        double m[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) m[r] = m0[r] + acc[r] * 1e-30;          // (a dependency on the previous repetition, numerically nil)
        sfor<0, 9>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            double piv;
            if constexpr (DPP) piv = dpp_bcast<i>(m[i]); else piv = bcast(m[i], i);
            const double ip = rcp_nr(piv);
            const double f = m[i] * ip;
            if constexpr (DPP) {
                const double nf = -f;
                sfor<i + 1, 9>([&](auto rr) { constexpr int r = decltype(rr)::value; dpp_update<i>(m[r], nf); });
            } else {
                sfor<i + 1, 9>([&](auto rr) { constexpr int r = decltype(rr)::value; m[r] = fma(-bcast(m[r], i), f, m[r]); });
            }
        });
#pragma unroll
        for (int r = 0; r < 9; ++r) acc[r] = m[r];
    }
    const long long t1 = clock64();
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
    if (blockIdx.x == 0) for (int r = 0; r < 9; ++r) out[lane * 9 + r] = acc[r];
}

template <int I> __device__ __forceinline__ void rcp_dpp(double m, double& r0, double& piv) {
    asm volatile("s_nop 1\n\tv_rcp_f64_dpp %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=&v"(r0), "=&v"(piv) : "v"(m));
}
__global__ void rcp_test(const double* in, double* out) {
    const int lane = threadIdx.x;
    double m = in[lane % 28 * 9] + 1.0 + lane;
    double r0, piv;
    rcp_dpp<3>(m, r0, piv);
    double ref = __builtin_amdgcn_rcp(piv);
    out[lane * 3] = r0; out[lane * 3 + 1] = ref; out[lane * 3 + 2] = piv;
}
int main() {
    // a 28-column augmented block: S = SPD 9x9 (columns 0..8), 19 more columns
    std::vector<double> h(28 * 9);
    for (int c = 0; c < 28; ++c) for (int r = 0; r < 9; ++r) h[c * 9 + r] = (c < 9 ? (c == r ? 12.0 + c : 0.3 / (1.0 + (c > r ? c - r : r - c))) : 0.1 * ((c * 7 + r * 3) % 11) - 0.4);
    double *din, *dout; long long* dclk;
    hipMalloc(&din, h.size() * 8); hipMalloc(&dout, 64 * 9 * 8); hipMalloc(&dclk, 1024 * 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    { double* o3; hipMalloc(&o3, 64 * 3 * 8); hipLaunchKernelGGL(rcp_test, dim3(1), dim3(64), 0, 0, din, o3); hipDeviceSynchronize();
      std::vector<double> h3(192); hipMemcpy(h3.data(), o3, 192 * 8, hipMemcpyDeviceToHost); int nb = 0; for (int l = 0; l < 64; ++l) nb += memcmp(&h3[l * 3], &h3[l * 3 + 1], 8) != 0;
      printf("v_rcp_f64_dpp vs v_rcp_f64 of the broadcast: %d of 64 lanes differ (lane 5: %.17g vs %.17g, piv %.17g)\n", nb, h3[15], h3[16], h3[17]); }
    const int reps = 2000;
    std::vector<double> oa(64 * 9), ob(64 * 9);
    for (int blocks : {1, 1024}) {
        for (int v = 0; v < 2; ++v) {
            for (int w = 0; w < 2; ++w) {
                if (v == 0) hipLaunchKernelGGL(pivots<false>, dim3(blocks), dim3(64), 0, 0, din, dout, dclk, reps);
                else hipLaunchKernelGGL(pivots<true>, dim3(blocks), dim3(64), 0, 0, din, dout, dclk, reps);
                hipDeviceSynchronize();
            }
            std::vector<long long> c(blocks);
            hipMemcpy(c.data(), dclk, blocks * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto x : c) s += (double)x;
            hipMemcpy(v == 0 ? oa.data() : ob.data(), dout, 64 * 9 * 8, hipMemcpyDeviceToHost);
            printf("%-9s %4d wave(s): %.0f clocks (s_memtime ticks) per node step\n", v == 0 ? "readlane" : "dpp", blocks, s / blocks / reps);
        }
    }
    // same bits: column c lives in lane c (A) / in its primary lane (B)
    int bad = 0;
    for (int c = 0; c < 28; ++c) {
        int lb = -1;
        for (int l = 0; l < 64 && lb < 0; ++l) { const int k = l >> 4, j = l & 15; const int cc = j < 9 ? j : (7 * k + j - 9 < 19 ? 9 + 7 * k + j - 9 : -1); if (cc == c) lb = l; }
        if (memcmp(&oa[c * 9], &ob[lb * 9], 72) != 0) ++bad;
    }
    printf("columns that differ between the two forms: %d of 28\n", bad);
    return bad != 0;
}
