// vr_math.h — transcendental kernels of the "exact" build.
//
// logf / expf / sinf / cosf written with +,-,*,/ (correctly rounded by default under hipcc),
// fmaf, rintf and integer bit operations only, so that the HIP kernels and the CPU oracle
// (which carries the same operation sequence, oracle/vr_oracle.c vro_shared_*) produce the
// same bits.  The reference calls its C library here (log: IntegratorDeltaTracking.cpp:65,
// exp: :88, cos/sin: Utils/DirectionSampler.cpp:62-63); accuracy of these kernels is within
// ~1-2 ulp of a correctly rounded result, checked in tests/test_shared_math.py.
//
// The "fast" build replaces them by the hardware v_log_f32 / v_exp_f32 / v_sin_f32 / v_cos_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vr {

__device__ __forceinline__ float vrm_logf(float x)
{
    uint32_t ix = __float_as_uint(x);
    ix += 0x3f800000u - 0x3f3504f3u;
    int k = (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    float f = __uint_as_float(ix) - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * fmaf(w, 0.24279078841f, 0.40000972152f);
    float t2 = z * fmaf(w, 0.28498786688f, 0.66666662693f);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    return fmaf(dk, 6.9313812256e-01f, (fmaf(s, hfsq + R, dk * 9.0580006145e-06f) - hfsq) + f);
}

__device__ __forceinline__ float vrm_expf(float x)
{
    if (!(x > -87.0f)) return 0.0f;
    if (x > 88.0f) return __builtin_huge_valf();
    float kf = rintf(x * 1.44269504089f);
    float r = fmaf(kf, -0.693145751953125f, x);
    r = fmaf(kf, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int k = (int)kf;
    return p * __uint_as_float((uint32_t)(k + 127) << 23);
}

__device__ __forceinline__ void vrm_sincos_reduce(float x, int& q, float& r)
{
    float qf = rintf(x * 0.636619772367581343f);
    float t = fmaf(qf, -1.5703125f, x);
    t = fmaf(qf, -4.837512969970703125e-4f, t);
    t = fmaf(qf, -7.54978995489188e-8f, t);
    q = (int)qf;
    r = t;
}
__device__ __forceinline__ float vrm_sin_poly(float r)
{
    float z = r * r;
    float p = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    p = fmaf(p, z, -1.6666654611e-1f);
    return fmaf(p * z, r, r);
}
__device__ __forceinline__ float vrm_cos_poly(float r)
{
    float z = r * r;
    float p = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    p = fmaf(p, z, 4.166664568298827e-2f);
    return fmaf(p, z * z, fmaf(z, -0.5f, 1.0f));
}
__device__ __forceinline__ void vrm_sincosf(float x, float& s, float& c)
{
    int q; float r;
    vrm_sincos_reduce(x, q, r);
    float sp = vrm_sin_poly(r), cp = vrm_cos_poly(r);
    float sv = (q & 1) ? cp : sp;
    float cv = (q & 1) ? sp : cp;
    s = (q & 2) ? -sv : sv;
    c = ((q + 1) & 2) ? -cv : cv;
}

} // namespace vr
