// Micro-benchmark 2: issue cost of the individual instructions of the tracking kernel's step loop (gfx950), each as
// inline assembly so that the compiler cannot fuse or drop it.  4 waves per SIMD, 16 independent chains per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu2 tools/ubench_valu2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
#define UNROLL 16
#define OP1(name, text)                                                                                                   \
    __global__ void __launch_bounds__(256) k_##name(uint32_t* out, uint32_t seed)                                         \
    {                                                                                                                     \
        uint32_t a[UNROLL];                                                                                               \
        uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x3f800000u;                                           \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) a[i] = 0x3f800000u + ((seed + threadIdx.x * 977u + i * 131u) & 0x7fffffu); \
        for (int it = 0; it < ITERS; ++it) {                                                                              \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) asm volatile(text : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s20", "s21"); \
        }                                                                                                                 \
        uint32_t r = 0;                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) r ^= a[i];                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                                   \
    }
OP1(add_f32, "v_add_f32 %0, %0, %1")
OP1(mul_f32, "v_mul_f32 %0, %0, %2")
OP1(fma_f32, "v_fma_f32 %0, %0, %2, %1")
OP1(fmamk, "v_fmamk_f32 %0, %0, 0x337fff58, %1")
OP1(add_u32, "v_add_u32 %0, %0, %1")
OP1(xor_b32, "v_xor_b32 %0, %0, %1")
OP1(lshr, "v_lshrrev_b32 %0, 8, %0")
OP1(mov, "v_mov_b32 %0, %1")
OP1(cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
OP1(cvt_flr, "v_cvt_flr_i32_f32 %0, %0")
OP1(floor, "v_floor_f32 %0, %0")
OP1(log, "v_log_f32 %0, %0")
OP1(exp, "v_exp_f32 %0, %0")
OP1(rcp, "v_rcp_f32 %0, %0")
OP1(med3, "v_med3_f32 %0, %0, %1, %2")
OP1(min3, "v_min3_f32 %0, %0, %1, %2")
OP1(max_f32, "v_max_f32 %0, %0, %1")
OP1(bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
OP1(alignbit, "v_alignbit_b32 %0, %0, %0, 21")
OP1(perm, "v_perm_b32 %0, %0, %1, %2")
OP1(cmp_vcc, "v_cmp_lt_f32 vcc, %0, %1")
OP1(cmp_sgpr, "v_cmp_lt_f32 s[20:21], %0, %1")
OP1(cmp_cndmask, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
OP1(cndmask_only, "v_cndmask_b32 %0, %0, %2, vcc")
OP1(mul_lo, "v_mul_lo_u32 %0, %0, %1")
OP1(mul_hi, "v_mul_hi_u32 %0, %0, %1")
OP1(mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
OP1(lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
OP1(and_or, "v_and_or_b32 %0, %0, %1, %2")
OP1(bfi, "v_bfi_b32 %0, %1, %0, %2")
OP1(salu_pair, "s_and_b64 s[20:21], vcc, exec\n v_add_f32 %0, %0, %1")
template <typename K> void run(const char* name, K kern, uint32_t* d, double insts)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;
    kern<<<blocks, 256>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(e0); kern<<<blocks, 256>>>(d, 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double stmts = 4.0 * ITERS * UNROLL;
    printf("%-16s %8.3f ms  %5.2f cycles per wave-statement per SIMD @2.4GHz (%g instr)\n", name, ms, ms * 1e6 / stmts * 2.4, insts);
}
#define RUN(name, n) run(#name, k_##name, d, n)
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 4 * 256 * 4);
    RUN(add_f32, 1); RUN(mul_f32, 1); RUN(fma_f32, 1); RUN(fmamk, 1); RUN(add_u32, 1); RUN(xor_b32, 1); RUN(lshr, 1); RUN(mov, 1);
    RUN(cvt_f32_u32, 1); RUN(cvt_flr, 1); RUN(floor, 1); RUN(log, 1); RUN(exp, 1); RUN(rcp, 1); RUN(med3, 1); RUN(min3, 1); RUN(max_f32, 1);
    RUN(bitop3, 1); RUN(alignbit, 1); RUN(perm, 1); RUN(cmp_vcc, 1); RUN(cmp_sgpr, 1); RUN(cmp_cndmask, 2); RUN(cndmask_only, 1);
    RUN(mul_lo, 1); RUN(mul_hi, 1); RUN(mad_u24, 1); RUN(lshl_or, 1); RUN(and_or, 1); RUN(bfi, 1); RUN(salu_pair, 2);
    return 0;
}
