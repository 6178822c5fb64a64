// Which packed-FP32 VALU instruction FORMS give a wrong result while ANOTHER kernel (matrix cores + packed FP32) runs on the same CUs?
// Registers only.  Every lane iterates a recurrence with the packed instruction and with its scalar equivalent and counts the iterations
// in which the bits differ, per form:
//   0: v_pk_fma_f32 (VGPR operands)                      1: v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (cross-half select)
//   2: v_pk_mul_f32 with an SGPR-pair operand            3: v_pk_add_f32 with an SGPR-pair operand, neg_lo / neg_hi
//   4: v_pk_mov_b32 op_sel:[1,0]                         5: v_pk_fma_f32 op_sel_hi:[1,0,1]
//   6: v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]         7: v_pk_fma_f32 op_sel:[1,0,0]
//   8: v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]         9: v_pk_add_f32 op_sel_hi:[0,1]
// build: hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared scripts/probes/pk_victim.hip -o islam_amd/lib/libislam_probe_pk.so
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define U(x) __float_as_uint(x)
__global__ __launch_bounds__(256) void pk_victim_kernel(unsigned* __restrict__ bad, int iters, float seed, f32x2 sc) {
    const int lane = threadIdx.x & 63;
    f32x2 x = {seed + 0.001f * lane, seed - 0.002f * lane};
    const f32x2 a = {0.99991f, 1.00003f}, b = {1e-3f, -2e-3f};
    unsigned nb[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        f32x2 y; float t0, t1;
        // 0
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(x), "v"(a), "v"(b));
        t0 = __builtin_fmaf(x.x, a.x, b.x); t1 = __builtin_fmaf(x.y, a.y, b.y);
        nb[0] += (U(y.x) != U(t0)) | (U(y.y) != U(t1));
        // 1: lo = x.lo + b.hi, hi = x.hi + b.lo
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(y) : "v"(x), "v"(b));
        nb[1] += (U(y.x) != U(x.x + b.y)) | (U(y.y) != U(x.y + b.x));
        // 2
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(y) : "s"(sc), "v"(x));
        nb[2] += (U(y.x) != U(sc.x * x.x)) | (U(y.y) != U(sc.y * x.y));
        // 3
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(y) : "v"(x), "s"(sc));
        nb[3] += (U(y.x) != U(x.x - sc.x)) | (U(y.y) != U(x.y - sc.y));
        // 4: lo = x.hi, hi = x.lo
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(y) : "v"(x), "v"(x));
        nb[4] += (U(y.x) != U(x.y)) | (U(y.y) != U(x.x));
        // 5: op_sel_hi:[1,0,1]: hi = x.hi * a.lo + b.hi
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(y) : "v"(x), "v"(a), "v"(b));
        nb[5] += (U(y.x) != U(__builtin_fmaf(x.x, a.x, b.x))) | (U(y.y) != U(__builtin_fmaf(x.y, a.x, b.y)));
        // 6: v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]: lo = x.lo * a.hi, hi = x.hi * a.lo
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(y) : "v"(x), "v"(a));
        nb[6] += (U(y.x) != U(x.x * a.y)) | (U(y.y) != U(x.y * a.x));
        // 7: v_pk_fma_f32 op_sel:[1,0,0]: lo = x.hi * a.lo + b.lo, hi = x.hi * a.hi + b.hi
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(y) : "v"(x), "v"(a), "v"(b));
        nb[7] += (U(y.x) != U(__builtin_fmaf(x.y, a.x, b.x))) | (U(y.y) != U(__builtin_fmaf(x.y, a.y, b.y)));
        // 8: v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]: lo = x.hi + b.lo, hi = x.lo + b.hi
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(y) : "v"(x), "v"(b));
        nb[8] += (U(y.x) != U(x.y + b.x)) | (U(y.y) != U(x.x + b.y));
        // 9: v_pk_add_f32 op_sel_hi:[0,1] (hi = x.lo + b.hi): no low-half select
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(y) : "v"(x), "v"(b));
        nb[9] += (U(y.x) != U(x.x + b.x)) | (U(y.y) != U(x.x + b.y));
        x.x = __builtin_fmaf(x.x, a.x, b.x); x.y = __builtin_fmaf(x.y, a.y, b.y);
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) if (nb[k]) atomicAdd(&bad[k * 4 + (lane >> 4)], nb[k]);
}
extern "C" int pk_victim_launch(unsigned* bad, int blocks, int iters, float seed, void* stream) {
    const f32x2 sc = {1.25f, 0.75f};
    hipLaunchKernelGGL(pk_victim_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bad, iters, seed, sc);
    return (int)hipGetLastError();
}
