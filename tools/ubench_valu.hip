// Micro-benchmark: issue cost of the VALU instructions the tracking kernel leans on (gfx950).
// Each kernel runs a dependent-free unrolled stream of one instruction kind on every SIMD with
// 4 waves per SIMD; prints cycles per wave-instruction per SIMD.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
#define UNROLL 16
typedef float float2v __attribute__((ext_vector_type(2)));
template <int OP> __global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed)
{
    uint32_t a[UNROLL]; float f[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + threadIdx.x * 977u + i * 131u; f[i] = 1.0f + (float)(a[i] & 1023) * 1e-3f; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) a[i] ^= a[i] >> 1 ^ seed;                       // 2 simple ops (xor+lshr fused? baseline)
            if (OP == 1) a[i] = a[i] * 0xD2511F53u;                      // v_mul_lo_u32
            if (OP == 2) a[i] = __umulhi(a[i], 0xD2511F53u);             // v_mul_hi_u32
            if (OP == 3) { unsigned long long p = (unsigned long long)a[i] * 0xD2511F53u; a[i] = (uint32_t)(p >> 32) ^ (uint32_t)p; } // mad_u64 + xor
            if (OP == 4) a[i] = __umul24(a[i], 0x1F53u) + 7u;            // v_mad_u32_u24
            if (OP == 5) f[i] = __builtin_amdgcn_rcpf(f[i]) + 0.5f;      // rcp + add
            if (OP == 6) f[i] = __expf(f[i]) * 0.3f;                     // mul + v_exp + mul
            if (OP == 7) f[i] = __logf(f[i]) + 2.0f;                     // v_log + mul + add
            if (OP == 8) f[i] = 1.0f / f[i] + 0.5f;                      // IEEE division + add
            if (OP == 9) f[i] = fmaf(f[i], 0.999f, 0.001f);              // v_fma
            if (OP == 10) f[i] = floorf(f[i] * 1.5f);                    // mul + floor
            if (OP == 11) f[i] = sqrtf(f[i]) + 1.0f;                     // IEEE sqrt + add
            if (OP == 15) f[i] = f[i] + 0.001f;                          // v_add_f32
            if (OP == 16) f[i] = fminf(fmaxf(f[i], 0.5f), 1.5f) * 1.0001f;   // v_med3 / min+max, then a mul
        }
        if (OP >= 12 && OP <= 14) {
            // packed fp32: two floats per lane and instruction.  Eight register pairs, one instruction each per pass
#pragma unroll
            for (int i = 0; i < UNROLL; i += 2) {
                float2v v = { f[i], f[i + 1] };
                const float2v c1 = { 0.999f, 0.998f }, c2 = { 0.001f, 0.002f };
                if (OP == 12) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(c1), "v"(c2));
                if (OP == 13) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(c1));
                if (OP == 14) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(c2));
                f[i] = v.x; f[i + 1] = v.y;
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) r ^= a[i] ^ __float_as_uint(f[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> void run(const char* name, uint32_t* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;   // 4 waves per SIMD
    k<OP><<<blocks, 256>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves * ITERS * UNROLL statements
    double stmts = 4.0 * ITERS * (OP >= 12 && OP <= 14 ? UNROLL / 2 : UNROLL);   // packed: one instruction per register pair
    printf("%-28s %8.3f ms  %6.2f ns per wave-statement per SIMD  (~%.1f cycles @2.4GHz)\n", name, ms, ms * 1e6 / stmts, ms * 1e6 / stmts * 2.4);
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0>("xor+shift+xor", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_hi_u32", d); run<3>("mad_u64_u32 + xor", d);
    run<4>("mad_u32_u24", d); run<5>("rcp + add", d); run<6>("expf (mul,exp,mul)", d); run<7>("logf (log,mul,add)", d);
    run<8>("IEEE div + add", d); run<9>("fma", d); run<10>("mul + floor", d); run<11>("IEEE sqrt + add", d);
    run<15>("v_add_f32", d); run<16>("clamp (med3) + mul", d);
    run<12>("v_pk_fma_f32 (2 floats)", d); run<13>("v_pk_mul_f32 (2 floats)", d); run<14>("v_pk_add_f32 (2 floats)", d);
    return 0;
}
