// How far apart must two MFMAs that accumulate into the SAME registers be for the matrix pipe of one SIMD to stay full?
// One wave per SIMD, v_mfma_f32_32x32x16_bf16, accumulator index sequences of period 12.  hipcc --offload-arch=gfx950 -O3 ...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MM(c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));

template <int KIND>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters) {
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    asm volatile("" : "+a"(a));
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { MM(c0) MM(c1) MM(c2) MM(c3) MM(c0) MM(c1) MM(c2) MM(c3) MM(c0) MM(c1) MM(c2) MM(c3) }          // distance 4
        if (KIND == 1) { MM(c0) MM(c0) MM(c0) MM(c0) MM(c0) MM(c0) MM(c0) MM(c0) MM(c0) MM(c0) MM(c0) MM(c0) }          // distance 1
        if (KIND == 2) { MM(c0) MM(c1) MM(c0) MM(c1) MM(c0) MM(c1) MM(c0) MM(c1) MM(c0) MM(c1) MM(c0) MM(c1) }          // distance 2
        if (KIND == 3) { MM(c0) MM(c1) MM(c2) MM(c0) MM(c1) MM(c2) MM(c0) MM(c1) MM(c2) MM(c0) MM(c1) MM(c2) }          // distance 3
        if (KIND == 4) { MM(c0) MM(c1) MM(c0) MM(c2) MM(c1) MM(c0) MM(c3) MM(c2) MM(c1) MM(c3) MM(c2) MM(c3) }          // conv_ws.hip's row order
        if (KIND == 5) { MM(c0) MM(c1) MM(c2) MM(c3) MM(c0) MM(c1) MM(c2) MM(c3) MM(c0) MM(c1) MM(c2) MM(c3) }          // (r outer, p inner)
    }
    const long long t1 = clock64();
    asm volatile("s_nop 15\n s_nop 15");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(float* out, long long* cyc, const char* name) {
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) { probe<KIND><<<256, 256>>>(out, cyc, iters); hipDeviceSynchronize(); }
    long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s %.1f shader clocks per MFMA\n", name, (double)h / (iters * 12.0));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<0>(out, cyc, "same accumulator every 4th");
    run<1>(out, cyc, "same accumulator back to back");
    run<2>(out, cyc, "same accumulator every 2nd");
    run<3>(out, cyc, "same accumulator every 3rd");
    run<4>(out, cyc, "conv_ws row order 0|10|210|321|32|3");
    return 0;
}
