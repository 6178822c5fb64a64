// Does ONE wave per SIMD hide its own VALU / LDS / accvgpr instructions behind its own MFMAs?  4 waves per workgroup (one per SIMD), one
// workgroup per CU (160 KB of LDS requested), a loop of {MFMA on one of four accumulators; K independent filler instructions}.
// Prints clocks per MFMA for K = 0 .. 12 and three fillers.  hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int K, int KIND>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters) {
    extern __shared__ float lds[];
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    float x[12];
    for (int i = 0; i < 12; ++i) x[i] = threadIdx.x * 0.5f + i;
    float y = 1.0001f;
    lds[threadIdx.x] = y;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define FILL(j) if (K > j) { if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[j]) : "v"(y)); \
                             else if (KIND == 1) asm volatile("v_accvgpr_write_b32 a200, %0" :: "v"(x[j])); \
                             else asm volatile("ds_read_b32 %0, %1" : "=v"(x[j]) : "v"((int)(threadIdx.x * 4))); }
#define STEP(cx) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(cx) : "v"(a), "v"(b)); \
        FILL(0) FILL(1) FILL(2) FILL(3) FILL(4) FILL(5) FILL(6) FILL(7) FILL(8) FILL(9) FILL(10) FILL(11)
        STEP(c0) STEP(c1) STEP(c2) STEP(c3) STEP(c0) STEP(c1) STEP(c2) STEP(c3)
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long t1 = clock64();
    asm volatile("s_nop 15\n s_nop 15");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 12; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int K, int KIND>
void run(float* out, long long* cyc, const char* name) {
    hipFuncSetAttribute((const void*)probe<K, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 2000;
    probe<K, KIND><<<256, 256, 150 * 1024>>>(out, cyc, iters);
    hipDeviceSynchronize();
    probe<K, KIND><<<256, 256, 150 * 1024>>>(out, cyc, iters);
    hipDeviceSynchronize();
    long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s K=%2d: %.1f shader clocks per MFMA\n", name, K, (double)h / (iters * 8.0));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
#define ALLK(KIND, name) run<0, KIND>(out, cyc, name); run<1, KIND>(out, cyc, name); run<2, KIND>(out, cyc, name); run<4, KIND>(out, cyc, name); \
    run<6, KIND>(out, cyc, name); run<8, KIND>(out, cyc, name); run<12, KIND>(out, cyc, name);
    ALLK(0, "v_fma_f32") ALLK(1, "v_accvgpr_write_b32") ALLK(2, "ds_read_b32")
    return 0;
}
