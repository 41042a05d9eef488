// vr_device.h — device-side building blocks of the integrator kernels (gfx950).
//
// Each block names the reference code whose results it must reproduce (paths relative to
// /root/reference/VolumeRenderer/RayTracer/).  The whole library is compiled with
// -ffp-contract=off: the only fused multiply-adds are the explicit fmaf calls, which are the
// ones NanoVDB itself makes (nanovdb/util/Ray.h:156-158, nanovdb/NanoVDB.h:1533-1535).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vr_math.h"

namespace vr {

#define VR_FLT_MAX 3.402823466e+38f
#define VR_PI_D 3.14159265358979323846

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 scale3(f3 a, float r) { return mk3(a.x * r, a.y * r, a.z * r); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // Utils.cpp:70-73
__device__ __forceinline__ f3 cross3(f3 a, f3 b)                                                  // Utils.cpp:4-11
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ f3 normalize3(f3 v)                                                    // Utils.cpp:18-27
{
    float mag2 = v.x * v.x + v.y * v.y + v.z * v.z;
    if (mag2 > 0) {
        float invMag = 1 / sqrtf(mag2);
        return mk3(v.x * invMag, v.y * invMag, v.z * invMag);
    }
    return v;
}
// Utils::clamp(lo, hi, v) = std::max(lo, std::min(hi, v))                                          Utils.cpp:75-78
__device__ __forceinline__ float clamp_lo_hi(float lo, float hi, float v)
{
    float m = (v < hi) ? v : hi;
    return (lo < m) ? m : lo;
}

// ---- math policy ------------------------------------------------------------------------
template <bool EXACT>
struct Math;
template <>
struct Math<true> {
    static __device__ __forceinline__ float log_(float x) { return vrm_logf(x); }
    static __device__ __forceinline__ float exp_(float x) { return vrm_expf(x); }
    static __device__ __forceinline__ void sincos_(float x, float& s, float& c) { vrm_sincosf(x, s, c); }
    static __device__ __forceinline__ float div_(float a, float b) { return a / b; }
    static __device__ __forceinline__ float rcp_(float x) { return 1 / x; }
    static __device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
    static __device__ __forceinline__ f3 normalize_(f3 v) { return normalize3(v); }
};
template <>
struct Math<false> {
    // bare v_log_f32 / v_exp_f32 (base 2): arguments here are normal floats, so the denormal
    // pre-scaling that __logf/__expf wrap around them (10+ instructions) is not needed
    static __device__ __forceinline__ float log_(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
    static __device__ __forceinline__ float exp_(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
    static __device__ __forceinline__ void sincos_(float x, float& s, float& c) { s = __sinf(x); c = __cosf(x); }
    static __device__ __forceinline__ float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
    // the event phases' divisions, square roots and normalisations: correctly rounded they are 10-15
    // instructions each and a third of an event phase; one-ulp hardware forms in the production mode
    static __device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ f3 normalize_(f3 v)
    {
        const float mag2 = v.x * v.x + v.y * v.y + v.z * v.z;
        const float inv = mag2 > 0 ? __builtin_amdgcn_rsqf(mag2) : 1.0f;
        return mk3(v.x * inv, v.y * inv, v.z * inv);
    }
};

// ---- random numbers: Philox4x32-10 keys the streams, xoshiro128+ runs them --------------------
// Replaces RandomGenerator (Utils/RandomGenerator.cpp:4-14): only the ranges of its draws are
// contract.  Layout (mirrored by the oracle's vro_rng):
//   Philox4x32-10 (Salmon et al. 2011), counter = (pixel, sample, tag, index), key = 64-bit seed;
//   pixel = (row << 16) | column in absolute image coordinates, so a pixel's streams do not depend
//   on crop, row partition or GPU count.
//   Trackers: one STREAM per walk — tag = scatter count so far ("bounce") for a camera walk,
//   | 0x80000000 for the shadow walk of that scatter event.  The Philox block (pixel, sample, tag, 0)
//   is the 128-bit state of a xoshiro128+ generator (Blackman & Vigna 2018: their generator for
//   floating-point numbers taken from the upper bits, which is the use here); the walk draws from it
//   in order.  An iteration that needs random numbers (sigma > 0, or the camera walk's roulette) takes
//   TWO words from ONE transition of the generator (xoshiro128p_next2: s0 + s3 and s1 + s2 of the same state):
//   first = free path, second = collision type (camera walk) or roulette (shadow walk), both
//   through their upper 24 bits; the camera walk's roulette at the loop top uses 16 bits from the low
//   bytes of the same two words.  Empty-space iterations draw nothing, so runs of them can be taken at
//   once.  The stream of bounce b+1 starts with the two words of the direction sample that ends bounce b.
//   Marchers: block (pixel, sample, 0, n) per march step n: word 0 jitter, word 1 roulette.
// The step loop is bound by its VALU instruction count (SQ_ACTIVE_INST_VALU = all SIMD cycles): a
// xoshiro128+ word is seven instructions, and Philox runs once per walk instead of once per two iterations.
__device__ __forceinline__ uint32_t xoshiro128p_next(uint32_t& s0, uint32_t& s1, uint32_t& s2, uint32_t& s3)
{
    const uint32_t result = s0 + s3;
    const uint32_t t = s1 << 9;
    const uint32_t x3 = s3 ^ s1;                                                   // s3 ^= s1
    const uint32_t n1 = __builtin_amdgcn_bitop3_b32(s1, s2, s0, 0x96);             // s2 ^= s0; s1 ^= s2
    const uint32_t n2 = __builtin_amdgcn_bitop3_b32(s2, s0, t, 0x96);              // s2 ^= s0; ...; s2 ^= t
    s0 ^= x3;                                                                      // s0 ^= s3
    s1 = n1; s2 = n2;
    s3 = __builtin_amdgcn_alignbit(x3, x3, 32 - 11);                               // s3 = rotl(s3, 11)
    return result;
}
// Two words from ONE state transition: the published output s0 + s3 and, from the two state words it does not read,
// s1 + s2.  Over the generator's period the four state words are jointly uniform, so the two sums are independent of
// each other; an iteration of a walk needs exactly two words (free path, collision type), and the second transition it
// used to pay for was seven of the ~40 VALU instructions of an iteration (+5 % on config 2: the step loop is bound by
// its instruction count).  The oracle draws the same way (rng_pair in oracle/vr_oracle.c).
__device__ __forceinline__ void xoshiro128p_next2(uint32_t& s0, uint32_t& s1, uint32_t& s2, uint32_t& s3, uint32_t& a, uint32_t& b)
{
    a = s0 + s3;
    b = s1 + s2;
    const uint32_t t = s1 << 9;
    const uint32_t x3 = s3 ^ s1;
    const uint32_t n1 = __builtin_amdgcn_bitop3_b32(s1, s2, s0, 0x96);
    const uint32_t n2 = __builtin_amdgcn_bitop3_b32(s2, s0, t, 0x96);
    s0 ^= x3;
    s1 = n1; s2 = n2;
    s3 = __builtin_amdgcn_alignbit(x3, x3, 32 - 11);
}
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3)
{
#ifndef VR_PHILOX_ROUNDS
#define VR_PHILOX_ROUNDS 10   // anything else is a timing experiment, never a shipped build
#endif
#pragma unroll
    for (int r = 0; r < VR_PHILOX_ROUNDS; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        // three-way xor in one v_bitop3_b32 (truth table 0x96)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}
// uniform in [0,1) with 24 bits, then RandomGenerator::getFloat(lo, hi) = lo + u*(hi-lo)
__device__ __forceinline__ float unit24(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float in_range(uint32_t w, float lo, float hi) { return (unit24(w) * (hi - lo)) + lo; }
// the camera walk's roulette draw: 16 bits in [0,1) from the low bytes of the iteration's two words
// (unit24 uses the upper 24 bits of each)
__device__ __forceinline__ float roulette16(uint32_t w_path, uint32_t w_event)
{
    return (float)(((w_path & 0xffu) << 8) | (w_event & 0xffu)) * (1.0f / 65536.0f);
}

// ---- nanovdb::Ray<float> ----------------------------------------------------------------------
struct Ray {
    f3    eye, dir;
    float t0, t1;
};
__device__ __forceinline__ f3 ray_at(f3 eye, f3 dir, float t)                                     // Ray.h:153-162
{
    return mk3(fmaf(t, dir.x, eye.x), fmaf(t, dir.y, eye.y), fmaf(t, dir.z, eye.z));
}
__device__ __forceinline__ f3 mat_mult(const float* m, f3 p)                                      // NanoVDB.h:1531-1536
{
    return mk3(fmaf(p.x, m[0], fmaf(p.y, m[1], p.z * m[2])), fmaf(p.x, m[3], fmaf(p.y, m[4], p.z * m[5])),
               fmaf(p.x, m[6], fmaf(p.y, m[7], p.z * m[8])));
}
// Ray::clip(CoordBBox): slab test against [min, max+1], same comparisons, so NaNs from 0*inf
// behave as in the reference                                                                   Ray.h:397-423,512-520
template <bool EXACT = true>
__device__ __forceinline__ bool ray_clip(f3 eye, f3 dir, float& t0, float& t1, const float* bmin, const float* bmax1)
{
    // mInvDir(1/dir)                                                                            Ray.h:69
    const float e[3] = { eye.x, eye.y, eye.z };
    const float inv[3] = { Math<EXACT>::rcp_(dir.x), Math<EXACT>::rcp_(dir.y), Math<EXACT>::rcp_(dir.z) };
    float a0 = t0, a1 = t1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float a = bmin[i], b = bmax1[i];
        if (a >= b) return false;
        a = (a - e[i]) * inv[i];
        b = (b - e[i]) * inv[i];
        if (a > b) { float tmp = a; a = b; b = tmp; }
        if (a > a0) a0 = a;
        if (b < a1) a1 = b;
        if (a0 > a1) return false;
    }
    t0 = a0; t1 = a1;
    return true;
}

// ---- density field: dense grid of 8^3 bricks ---------------------------------------------------
// Replaces ReadAccessor<float,0,1,2>::getValue(Coord::Floor(p)) (NanoVDB.h:5248-5263, :1133,
// :818-821): nearest voxel by floor, background outside the bbox.  Brick b = (bi*nby + bj)*nbz + bk
// holds voxel (i,j,k) at b*512 + ((i&7)<<6 | (j&7)<<3 | (k&7)) — NanoVDB's own leaf order.
struct GridDev {
    const float* bricks;
    int32_t      min[3];
    int32_t      dim[3]; // bbox extent in voxels (max - min + 1)
    int32_t      nb[3];  // bricks per axis
    float        background;
};
// position of voxel (i & 7, j & 7, k & 7) inside its brick, and back.  -DVR_BRICK_MORTON is an experiment: a
// 64-byte line then holds 2 x 2 x 4 voxels instead of 1 x 2 x 8.
__host__ __device__ __forceinline__ uint32_t brick_local(uint32_t i, uint32_t j, uint32_t k)
{
#ifdef VR_BRICK_MORTON
    return ((i & 6u) << 6) | ((j & 6u) << 4) | ((k & 4u) << 2) | ((i & 1u) << 3) | ((j & 1u) << 2) | (k & 3u);
#else
    return ((i & 7u) << 6) | ((j & 7u) << 3) | (k & 7u);
#endif
}
__host__ __device__ __forceinline__ void brick_local_decode(uint32_t local, int& i, int& j, int& k)
{
#ifdef VR_BRICK_MORTON
    i = (int)(((local >> 6) & 6u) | ((local >> 3) & 1u)); j = (int)(((local >> 4) & 6u) | ((local >> 2) & 1u)); k = (int)(((local >> 2) & 4u) | (local & 3u));
#else
    i = (int)(local >> 6); j = (int)((local >> 3) & 7u); k = (int)(local & 7u);
#endif
}
__device__ __forceinline__ uint32_t brick_offset(const GridDev& g, uint32_t i, uint32_t j, uint32_t k)
{
    uint32_t b = ((i >> 3) * (uint32_t)g.nb[1] + (j >> 3)) * (uint32_t)g.nb[2] + (k >> 3);
    return (b << 9) | brick_local(i, j, k);
}

// ---- runs of empty-space steps ---------------------------------------------------------------
// In a voxel with sigma <= 0 the reference's walks only do `tFar += step` with a constant step and
// no random draw (IntegratorRatioTracking.cpp:84-96, :251-263; IntegratorDeltaTracking.cpp:70-82).
// empty_run takes as many of those iterations at once as START inside a box — the current voxel, or
// the cube of empty voxels around it that the skip grid reports; the caller passes t_safe =
// min(exit parameter of the shrunk box, t1) — and returns their number (>= 1).  All but the last
// also END inside the box (and at t <= t1, so none of them is the walk's last); the last one is an
// ordinary float addition that may leave it.  t advances exactly as that many sequential float
// additions would:  inside one binade [2^e, 2^(e+1)) a float is m * 2^(e-23) with integer m, and
// fl(t + step) = t + q * 2^(e-23) with q = step / 2^(e-23) rounded to nearest (ties are not
// handled here: they fall back to a single step), so k additions add k * q * 2^(e-23), exactly.  The box is shrunk by `slack` on every side, which covers the rounding of
// fmaf(t, dir, eye) and of the exit-distance estimate; estimates only ever make the run shorter.
__device__ __forceinline__ uint32_t empty_run(float& t, float t_safe, float step)
{
    // q * ulp(t), read off the first step itself: fl(t + step) - t is exact while the sum stays in t's binade (and
    // where it does not, the limit below leaves nothing to batch)
    const float    qu = (t + step) - t;
    const uint32_t tb = __float_as_uint(t);                        // t > 0: sign bit clear
    const float    half_ulp = __uint_as_float(tb & 0x7f800000u) * 5.9604644775390625e-08f;   // 2^(e-24); 0 for a denormal t: no batch
    // a tie (step / ulp(t) = n + 1/2) rounds to even and alternates between n and n + 1: not batched.  The difference
    // is exact (qu and step are within a factor of two of each other once step >= ulp / 2)
    const bool     regular = qu > 0.0f && fabsf(qu - step) != half_ulp;
    // the batched steps stay at or below the last float of t's binade, so every one of them adds exactly qu, and at
    // or below t_safe
    const float t_last = __uint_as_float(tb | 0x7fffffu);
    const float t_lim = t_safe < t_last ? t_safe : t_last;
    const float kf = floorf((t_lim - t) * __builtin_amdgcn_rcpf(qu) * 0.999999f) - 1.0f;
    uint32_t k = 0;
    if (regular && kf >= 1.0f) {
        k = (uint32_t)kf;
        t = fmaf(kf, qu, t);   // exact: k q < 2^23 and the sum does not leave the binade
    }
    // the k steps above all START and END inside the box; one more ordinary step starts inside it
    // (same sigma <= 0) and may end anywhere: the caller looks the new position up as usual
    t += step;
    return k + 1;
}

// (int32_t)floorf(x) in one instruction (v_cvt_flr_i32_f32; the compiler emits v_floor + v_cvt_i32).  Same
// result wherever the floor fits an int32, and coordinates are clipped to the bbox long before that.
__device__ __forceinline__ int32_t floor_to_int(float x)
{
    int32_t r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ float grid_value_nocache(const GridDev& g, f3 p)
{
    int32_t i = floor_to_int(p.x) - g.min[0];
    int32_t j = floor_to_int(p.y) - g.min[1];
    int32_t k = floor_to_int(p.z) - g.min[2];
    if ((uint32_t)i >= (uint32_t)g.dim[0] || (uint32_t)j >= (uint32_t)g.dim[1] || (uint32_t)k >= (uint32_t)g.dim[2])
        return g.background;
    return g.bricks[brick_offset(g, (uint32_t)i, (uint32_t)j, (uint32_t)k)];
}

// ---- Perlin noise (Utils/PerlinNoiseSampler.cpp) --------------------------------------------------
// Ken Perlin's reference permutation — the table of Utils/PerlinNoiseSampler.cpp:11-28 (data)
__constant__ uint8_t c_perm[256] = {
    151, 160, 137, 91,  90,  15,  131, 13,  201, 95,  96,  53,  194, 233, 7,   225, 140, 36,  103, 30,  69,  142, 8,   99,
    37,  240, 21,  10,  23,  190, 6,   148, 247, 120, 234, 75,  0,   26,  197, 62,  94,  252, 219, 203, 117, 35,  11,  32,
    57,  177, 33,  88,  237, 149, 56,  87,  174, 20,  125, 136, 171, 168, 68,  175, 74,  165, 71,  134, 139, 48,  27,  166,
    77,  146, 158, 231, 83,  111, 229, 122, 60,  211, 133, 230, 220, 105, 92,  41,  55,  46,  245, 40,  244, 102, 143, 54,
    65,  25,  63,  161, 1,   216, 80,  73,  209, 76,  132, 187, 208, 89,  18,  169, 200, 196, 135, 130, 116, 188, 159, 86,
    164, 100, 109, 198, 173, 186, 3,   64,  52,  217, 226, 250, 124, 123, 5,   202, 38,  147, 118, 126, 255, 82,  85,  212,
    207, 206, 59,  227, 47,  16,  58,  17,  182, 189, 28,  42,  223, 183, 170, 213, 119, 248, 152, 2,   44,  154, 163, 70,
    221, 153, 101, 155, 167, 43,  172, 9,   129, 22,  39,  253, 19,  98,  108, 110, 79,  113, 224, 232, 178, 185, 112, 104,
    218, 246, 97,  228, 251, 34,  242, 193, 238, 210, 144, 12,  191, 179, 162, 241, 81,  51,  145, 235, 249, 14,  239, 107,
    49,  192, 214, 31,  181, 199, 106, 157, 184, 84,  204, 176, 115, 121, 50,  45,  127, 4,   150, 254, 138, 236, 205, 93,
    222, 114, 67,  29,  24,  72,  243, 141, 128, 195, 78,  66,  215, 61,  156, 180
};
__device__ __forceinline__ int pp(int i) { return c_perm[i & 255]; }
__device__ __forceinline__ float pn_fade(float t) { return t * t * t * (t * (t * 6 - 15) + 10); }        // :32
__device__ __forceinline__ float pn_lerp(float t, float a, float b) { return a + t * (b - a); }            // :33
__device__ __forceinline__ float pn_grad(int hash, float x, float y, float z)                              // :34-40
{
    int   h = hash & 15;
    float u = h < 8 ? x : y;
    float v = h < 4 ? y : (h == 12 || h == 14) ? x : z;
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}
__device__ __forceinline__ float perlin_noise(float x, float y, float z)                                                   // :46-68
{
    int X = (int)floorf(x) & 255, Y = (int)floorf(y) & 255, Z = (int)floorf(z) & 255;
    x -= floorf(x); y -= floorf(y); z -= floorf(z);
    float u = pn_fade(x), v = pn_fade(y), w = pn_fade(z);
    int A = pp(X) + Y, AA = pp(A) + Z, AB = pp(A + 1) + Z;
    int B = pp(X + 1) + Y, BA = pp(B) + Z, BB = pp(B + 1) + Z;
    return pn_lerp(w,
        pn_lerp(v, pn_lerp(u, pn_grad(pp(AA), x, y, z), pn_grad(pp(BA), x - 1, y, z)),
                   pn_lerp(u, pn_grad(pp(AB), x, y - 1, z), pn_grad(pp(BB), x - 1, y - 1, z))),
        pn_lerp(v, pn_lerp(u, pn_grad(pp(AA + 1), x, y, z - 1), pn_grad(pp(BA + 1), x - 1, y, z - 1)),
                   pn_lerp(u, pn_grad(pp(AB + 1), x, y - 1, z - 1), pn_grad(pp(BB + 1), x - 1, y - 1, z - 1))));
}
// PerlinNoiseSampler::eval_density: noise at frequency 1.8 remapped to [0,1]                  PerlinNoiseSampler.cpp:94-98
__device__ __forceinline__ float perlin_eval_density(f3 p)
{
    const float freq = 1.8f;
    return (float)((1 + perlin_noise(p.x * freq, p.y * freq, p.z * freq)) * 0.5);
}

// ---- phase function and direction sampling -----------------------------------------------------
// PhaseFunction::henyey_greenstein                                                              PhaseFunction.cpp:4-8
template <bool EXACT = true>
__device__ __forceinline__ float hg_phase(float g, float cos_theta)
{
    float denom = 1 + g * g - 2 * g * cos_theta;
    if (!EXACT) return (0.07957747154594767f * (1 - g * g)) * __builtin_amdgcn_rcpf(denom * __builtin_amdgcn_sqrtf(denom));
    return (float)(1 / (4 * VR_PI_D) * (1 - g * g) / (denom * sqrtf(denom)));
}
// DirectionSampler::sampleHenyeyGreenstein + ONB::Update                       DirectionSampler.cpp:38-72, ONB.cpp:5-22
template <bool EXACT>
__device__ __forceinline__ f3 hg_sample(float g, f3 direction, float xi, float xi2)
{
    f3 n = direction, s;
    if (fabsf(n.x) > fabsf(n.z)) { s.x = -n.y; s.y = n.x; s.z = 0; }
    else                         { s.x = 0; s.y = -n.z; s.z = n.y; }
    s = Math<EXACT>::normalize_(s);
    f3 t = cross3(n, s);
    f3 nd = Math<EXACT>::normalize_(direction);
    float cos_theta;
    if (g != 0.0f) {
        float aux = Math<EXACT>::div_((1 - g * g), (1 - g + 2 * g * xi));
        cos_theta = Math<EXACT>::div_((1 + g * g - (aux * aux)), (2 * g));
    } else {
        cos_theta = (float)(1.0 - 2.0 * (double)xi);
    }
    float c2 = 1.0f - cos_theta * cos_theta;
    float sin_theta = Math<EXACT>::sqrt_((0.0f < c2) ? c2 : 0.0f);
    float phi = EXACT ? (float)((double)(xi2 * 2) * VR_PI_D) : xi2 * 6.283185307179586f;
    float sp, cp;
    Math<EXACT>::sincos_(phi, sp, cp);
    f3 v1 = scale3(s, sin_theta * cp);
    f3 v2 = scale3(t, sin_theta * sp);
    f3 v3 = scale3(nd, cos_theta);
    return add3(add3(v1, v2), v3);
}

} // namespace vr
