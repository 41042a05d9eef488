// Micro-benchmark 3: throughput of the scalar unit (ONE per CU, shared by its four SIMDs) beside the vector ALUs (gfx950).
// A divergent `if` costs s_and_saveexec / s_or_b64 exec / s_cbranch_execz: the tracking kernel's trip is 310 VALU and
// 230 scalar instructions.  Each kernel runs WAVES waves per SIMD on every CU; a statement = `nv` vector and `ns` scalar
// instructions, 16 statements per loop body.  Printed: CU cycles per statement per wave-set, i.e. how many cycles the CU needs
// to run one statement for ONE wave on EACH of its four SIMDs (4 wave-statements).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_salu tools/ubench_salu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
#define UNROLL 16
#define OPK(name, text)                                                                                                   \
    __global__ void __launch_bounds__(256) k_##name(uint32_t* out, uint32_t seed)                                         \
    {                                                                                                                     \
        uint32_t a[UNROLL];                                                                                               \
        uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x3f800000u;                                           \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) a[i] = 0x3f800000u + ((seed + threadIdx.x * 977u + i * 131u) & 0x7fffffu); \
        for (int it = 0; it < ITERS; ++it) {                                                                              \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) asm volatile(text : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25"); \
        }                                                                                                                 \
        uint32_t r = 0;                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) r ^= a[i];                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                                   \
    }
OPK(v1, "v_add_f32 %0, %0, %1")
OPK(s1, "s_and_b64 s[20:21], s[22:23], exec")
OPK(s2, "s_and_b64 s[20:21], s[22:23], exec\n s_or_b64 s[24:25], s[22:23], exec")
OPK(v1s1, "v_add_f32 %0, %0, %1\n s_and_b64 s[20:21], s[22:23], exec")
OPK(v2s1, "v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2\n s_and_b64 s[20:21], s[22:23], exec")
OPK(v4s3, "v_add_f32 %0, %0, %1\n s_and_b64 s[20:21], s[22:23], exec\n v_mul_f32 %0, %0, %2\n s_or_b64 s[24:25], s[22:23], exec\n v_add_f32 %0, %0, %1\n s_andn2_b64 s[20:21], s[24:25], exec\n v_mul_f32 %0, %0, %2")
OPK(v4, "v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2")
OPK(v4cmp2, "v_add_f32 %0, %0, %1\n v_cmp_lt_f32 s[20:21], %0, %1\n v_mul_f32 %0, %0, %2\n v_cmp_lt_f32 s[22:23], %0, %2")
OPK(v4cmp2s3, "v_add_f32 %0, %0, %1\n v_cmp_lt_f32 s[20:21], %0, %1\n s_and_b64 s[24:25], s[20:21], exec\n v_mul_f32 %0, %0, %2\n v_cmp_lt_f32 s[22:23], %0, %2\n s_or_b64 s[24:25], s[22:23], s[24:25]\n s_andn2_b64 s[20:21], s[24:25], exec")
OPK(saveexec, "v_cmp_lt_f32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]")
OPK(cndmask, "v_cmp_lt_f32 vcc, %0, %1\n v_add_f32 %0, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
template <typename K> void run(const char* name, K kern, uint32_t* d, int waves, int nv, int ns)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves;       // one 256-thread block = one wave on each SIMD of a CU
    kern<<<blocks, 256>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(e0); kern<<<blocks, 256>>>(d, 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double stmts = (double)waves * ITERS * UNROLL;     // statements per SIMD (and per CU: wave-sets)
    const double cyc = ms * 1e-3 * 2.4e9 / stmts;
    printf("%-10s %d waves/SIMD  %2d VALU + %d SALU per statement: %6.2f CU cycles per statement on all four SIMDs -> VALU %.2f, SALU %.2f per CU cycle\n",
           name, waves, nv, ns, cyc, 4.0 * nv / cyc, 4.0 * ns / cyc);
}
#define RUN(name, w, nv, ns) run(#name, k_##name, d, w, nv, ns)
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 4, 6}) {
        RUN(v1, w, 1, 0); RUN(s1, w, 0, 1); RUN(s2, w, 0, 2); RUN(v1s1, w, 1, 1); RUN(v2s1, w, 2, 1); RUN(v4, w, 4, 0); RUN(v4s3, w, 4, 3);
        RUN(v4cmp2, w, 4, 0); RUN(v4cmp2s3, w, 4, 3); RUN(saveexec, w, 2, 2); RUN(cndmask, w, 3, 0);
    }
    return 0;
}
