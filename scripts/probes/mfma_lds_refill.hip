// conv_ws.hip's multiply step in isolation: 12 MFMAs in row groups 1|2|3|3|2|1, behind every group one ds_read_b128 that refills the
// operand register the group has just left, s_waitcnt lgkmcnt(5) in front of every group.  One wave per SIMD, four waves per CU.
// Variants: no reads; reads at pixel stride 272 B (conv_ws.hip); reads but no waits; b64 reads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MM(c, bq) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(bq));
#define RD(bq, off) if (KIND >= 1) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(bq) : "v"(addr));
#define WT() if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(5)");

template <int KIND>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters) {
    extern __shared__ unsigned short lds[];
    for (int i = threadIdx.x; i < 60000; i += 256) lds[i] = (unsigned short)i;
    __syncthreads();
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8};
    asm volatile("" : "+a"(a));
    bf16x8 b0 = a, b1 = a, b2 = a, b3 = a, b4 = a, b5 = a;
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const int lane = threadIdx.x & 63, li = lane & 31, kg = lane >> 5;
    const int addr = li * 272 + kg * 16;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        WT() MM(c0, b0) RD(b0, 0)
        WT() MM(c1, b1) MM(c0, b1) RD(b1, 9248)
        WT() MM(c2, b2) MM(c1, b2) MM(c0, b2) RD(b2, 18496)
        WT() MM(c3, b3) MM(c2, b3) MM(c1, b3) RD(b3, 27744)
        WT() MM(c3, b4) MM(c2, b4) RD(b4, 36992)
        WT() MM(c3, b5) RD(b5, 46240)
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const long long t1 = clock64();
    asm volatile("s_nop 15\n s_nop 15");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + b0[0] + b1[0] + b2[0] + b3[0] + b4[0] + b5[0];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(float* out, long long* cyc, const char* name) {
    hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) { probe<KIND><<<256, 256, 150 * 1024>>>(out, cyc, iters); hipDeviceSynchronize(); }
    long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-46s %.1f shader clocks per MFMA\n", name, (double)h / (iters * 12.0));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<0>(out, cyc, "MFMAs only");
    run<1>(out, cyc, "refill reads + lgkmcnt(5) waits (conv_ws.hip)");
    run<2>(out, cyc, "refill reads, no waits (wrong data, timing only)");
    return 0;
}
