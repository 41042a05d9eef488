/*
 * vr_oracle.c — CPU oracle (TEST INFRASTRUCTURE, see vr_oracle.h for the pinning status).
 *
 * Plain-C restatement of the reference hot path.  "REF:" comments give the file:line in
 * /root/reference/VolumeRenderer/RayTracer/ that each block follows.  Float/double mixing,
 * operand order and comparison direction follow the reference expression by expression;
 * build with -ffp-contract=off (oracle/Makefile) so no product-sum is fused that the
 * reference does not fuse (the explicit fmaf calls are the reference's own, from NanoVDB).
 */
#include "vr_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------ */
/* transcendental back-ends                                                             */
/* ------------------------------------------------------------------------------------ */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Shared polynomial kernels.  The HIP "exact" build carries the same operation sequence
 * (volume_renderer_amd/csrc/vr_math.h); only +,-,*,/ (correctly rounded on both sides),
 * fmaf, rintf and integer bit operations are used, so the results are bit-identical. */
float vro_shared_logf(float x)
{
    /* x > 0, finite, normal.  x = m * 2^k with m in [sqrt(1/2), sqrt(2)). */
    uint32_t ix = f2u(x);
    ix += 0x3f800000u - 0x3f3504f3u;
    int k = (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    float f = u2f(ix) - 1.0f;
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

float vro_shared_expf(float x)
{
    if (!(x > -87.0f)) return 0.0f;
    if (x > 88.0f) return INFINITY;
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
    return p * u2f((uint32_t)(k + 127) << 23);
}

/* sin/cos for |x| <= ~8 (the path only needs [0, 2*pi)) */
static inline void shared_sincos_reduce(float x, int* q, float* r)
{
    float qf = rintf(x * 0.636619772367581343f);
    float t = fmaf(qf, -1.5703125f, x);
    t = fmaf(qf, -4.837512969970703125e-4f, t);
    t = fmaf(qf, -7.54978995489188e-8f, t);
    *q = (int)qf;
    *r = t;
}
static inline float shared_sin_poly(float r)
{
    float z = r * r;
    float p = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    p = fmaf(p, z, -1.6666654611e-1f);
    return fmaf(p * z, r, r);
}
static inline float shared_cos_poly(float r)
{
    float z = r * r;
    float p = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    p = fmaf(p, z, 4.166664568298827e-2f);
    return fmaf(p, z * z, fmaf(z, -0.5f, 1.0f));
}
float vro_shared_sinf(float x)
{
    int q; float r;
    shared_sincos_reduce(x, &q, &r);
    float v = (q & 1) ? shared_cos_poly(r) : shared_sin_poly(r);
    return (q & 2) ? -v : v;
}
float vro_shared_cosf(float x)
{
    int q; float r;
    shared_sincos_reduce(x, &q, &r);
    float v = (q & 1) ? shared_sin_poly(r) : shared_cos_poly(r);
    return ((q + 1) & 2) ? -v : v;
}

static int g_math_mode = VRO_MATH_LIBM;
void vro_set_math_mode(int mode) { g_math_mode = mode; }
int  vro_get_math_mode(void) { return g_math_mode; }

static inline float m_logf(float x) { return g_math_mode ? vro_shared_logf(x) : logf(x); }
static inline float m_expf(float x) { return g_math_mode ? vro_shared_expf(x) : expf(x); }


/* ------------------------------------------------------------------------------------ */
/* random numbers                                                                       */
/* ------------------------------------------------------------------------------------ */

/* Philox4x32-10 (Salmon et al., SC'11; Random123 known-answer vectors in tests/).  No
 * reference counterpart: the reference's RNG is REF Utils/RandomGenerator.cpp:4-14; only
 * the RANGES of its draws are contract. */
void vro_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* std::mt19937 (REF Utils/RandomGenerator.cpp:6).  state625[624] is the index. */
void vro_mt19937_seed(uint32_t* mt, uint32_t seed)
{
    mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    mt[624] = 624;
}
uint32_t vro_mt19937_next(uint32_t* mt)
{
    if (mt[624] >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = mt[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            mt[i] = v;
        }
        mt[624] = 0;
    }
    uint32_t y = mt[mt[624]++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
/* RandomGenerator::getFloat (REF Utils/RandomGenerator.cpp:9-14) with libstdc++'s
 * uniform_real_distribution<float>: generate_canonical<float,24> draws one 32-bit word,
 * (float(word) / 2^32), clamps 1.0 to nextafter(1,0), then a + u*(b-a). */
float vro_reference_get_float(uint32_t* mt, float lo, float hi)
{
    float sum = (float)vro_mt19937_next(mt);
    float u = sum / 4294967296.0f;
    if (u >= 1.0f) u = nextafterf(1.0f, 0.0f);
    return (u * (hi - lo)) + lo;
}
/* `static std::default_random_engine e; static std::uniform_real_distribution<> dis(0,1)`
 * (REF integrators/IntegratorHomogeneousRayMarcherImproved.cpp:82-84): libstdc++'s
 * default_random_engine is minstd_rand0; generate_canonical<double,53> draws twice. */
void vro_minstd_canonical_double(uint32_t* state, double* out)
{
    const double r = 2147483646.0;
    uint32_t x1 = (uint32_t)(((uint64_t)(*state) * 16807u) % 2147483647u);
    uint32_t x2 = (uint32_t)(((uint64_t)x1 * 16807u) % 2147483647u);
    *state = x2;
    double sum = (double)(x1 - 1u) + (double)(x2 - 1u) * r;
    double ret = sum / (r * r);
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    *out = ret;
}

/* Random-number layout (the GPU path's; no reference counterpart):
 *   Philox4x32-10, counter = (pixel, sample, tag, index), key = 64-bit seed;
 *   pixel = (row << 16) | column (absolute).
 *   Trackers: one STREAM per walk: tag = scatter count so far for a camera walk, | 0x80000000 for the
 *   shadow walk of that scatter event.  The Philox block (pixel, sample, tag, 0) is the state of a
 *   xoshiro128+ generator (Blackman & Vigna 2018, public domain: their generator for floats taken from
 *   the upper bits) from which the walk draws in order.
 *   An iteration that needs random numbers takes TWO words at its first draw, both from ONE transition of the
 *   generator (vro_xoshiro128p_next2: s0 + s3 and s1 + s2 of the same state): first = free path,
 *   second = collision type (camera walk) or roulette (shadow walk), both through their upper 24
 *   bits; the camera walk's Russian roulette at the loop top takes 16 bits from the low bytes of the
 *   same two words.  Iterations in empty space (sigma <= 0, no roulette) draw nothing, which lets the
 *   GPU take runs of them at once.  The stream of bounce b+1 starts with the two words of the
 *   direction sample that ends bounce b.
 *   Marchers: block (pixel, sample, 0, n) per march step n: word 0 = jitter, word 1 = roulette.
 * In VRO_RNG_REFERENCE mode every draw is the next RandomGenerator::getFloat of the chunk's
 * mt19937, in the reference's call order. */
typedef struct vro_rng {
    int      mode;
    uint32_t key[2];
    uint32_t pixel, sample, tag, step, cur;
    uint32_t blk[4], blk_tag, blk_index;
    int      blk_valid;
    uint32_t s[4];        /* xoshiro128+ state of the current walk */
    uint32_t pair[2];     /* the current iteration's two words */
    int      pair_valid;
    uint32_t saved_tag, saved_step, saved_s[4];
    uint32_t* mt;         /* 625 words, owned by the row chunk */
    uint32_t* jitter;     /* minstd state, shared (the reference's function-static engine) */
} vro_rng;

static inline float unit24(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

static const uint32_t* rng_block(vro_rng* g, uint32_t tag, uint32_t index)
{
    if (!g->blk_valid || g->blk_tag != tag || g->blk_index != index) {
        uint32_t ctr[4] = { g->pixel, g->sample, tag, index };
        vro_philox4x32_10(ctr, g->key, g->blk);
        g->blk_tag = tag; g->blk_index = index;
        g->blk_valid = 1;
    }
    return g->blk;
}
/* xoshiro128+ 1.0 as published: result = s0 + s3, then the xorshift/rotate state update */
uint32_t vro_xoshiro128p_next(uint32_t s[4])
{
    const uint32_t result = s[0] + s[3];
    const uint32_t t = s[1] << 9;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = (s[3] << 11) | (s[3] >> 21);
    return result;
}
/* Two words from ONE transition: the published output s0 + s3 and s1 + s2 of the same state (the two state words the
 * published output does not read; over the period the four words are jointly uniform, so the sums are independent). */
void vro_xoshiro128p_next2(uint32_t s[4], uint32_t out[2])
{
    out[1] = s[1] + s[2];
    out[0] = vro_xoshiro128p_next(s);
}
/* the stream of walk `tag`: its state is the Philox block (pixel, sample, tag, 0) */
static void rng_seed_stream(vro_rng* g)
{
    uint32_t ctr[4] = { g->pixel, g->sample, g->tag, 0u };
    vro_philox4x32_10(ctr, g->key, g->s);
    if ((g->s[0] | g->s[1] | g->s[2] | g->s[3]) == 0u) g->s[0] = 1u;
    g->pair_valid = 0;
}
static const uint32_t* rng_pair(vro_rng* g)
{
    if (!g->pair_valid) {
        vro_xoshiro128p_next2(g->s, g->pair);
        g->pair_valid = 1;
    }
    return g->pair;
}
static void rng_begin_path(vro_rng* g, uint32_t i, uint32_t j, uint32_t s)
{
    g->pixel = (j << 16) | i;
    g->sample = s;
    g->tag = 0;
    g->step = 0;
    g->blk_valid = 0;
    rng_seed_stream(g);
}
/* top of every loop iteration of a walk */
static void rng_begin_step(vro_rng* g)
{
    g->cur = g->step++;
    g->pair_valid = 0;
}
/* getFloat(0, 1) of the roulette at the loop top (REF IntegratorRatioTracking.cpp:60) */
static float rng_roulette(vro_rng* g)
{
    if (g->mode == VRO_RNG_REFERENCE) return vro_reference_get_float(g->mt, 0.0f, 1.0f);
    const uint32_t* w = rng_pair(g);
    return (float)(((w[0] & 0xffu) << 8) | (w[1] & 0xffu)) * (1.0f / 65536.0f);
}
/* getFloat(0.00001f, 1.0f) of the free-path sample (REF :80, :247) */
static float rng_free_path(vro_rng* g)
{
    if (g->mode == VRO_RNG_REFERENCE) return vro_reference_get_float(g->mt, 0.00001f, 1.0f);
    return (unit24(rng_pair(g)[0]) * (1.0f - 0.00001f)) + 0.00001f;
}
/* getFloat(0, 1) of the collision type (REF :108) / of the shadow walk's roulette (REF :275) */
static float rng_event(vro_rng* g)
{
    if (g->mode == VRO_RNG_REFERENCE) return vro_reference_get_float(g->mt, 0.0f, 1.0f);
    return (unit24(rng_pair(g)[1]) * (1.0f - 0.0f)) + 0.0f;
}
/* the two getFloat(0, 0.9999f) of sampleHenyeyGreenstein (REF Utils/DirectionSampler.cpp:43,58):
 * the first two words of the NEXT bounce's stream, which starts here */
static void rng_direction(vro_rng* g, float* xi1, float* xi2)
{
    if (g->mode == VRO_RNG_REFERENCE) {
        *xi1 = vro_reference_get_float(g->mt, 0.0f, 0.9999f);
        *xi2 = vro_reference_get_float(g->mt, 0.0f, 0.9999f);
        return;
    }
    g->tag += 1;
    g->step = 0;
    rng_seed_stream(g);
    uint32_t w[2];
    vro_xoshiro128p_next2(g->s, w);
    *xi1 = (unit24(w[0]) * (0.9999f - 0.0f)) + 0.0f;
    *xi2 = (unit24(w[1]) * (0.9999f - 0.0f)) + 0.0f;
}
/* the shadow walk of bounce b draws from stream (b | 0x80000000) */
static void rng_push_shadow(vro_rng* g)
{
    g->saved_tag = g->tag;
    g->saved_step = g->step;
    memcpy(g->saved_s, g->s, sizeof g->s);
    g->tag |= 0x80000000u;
    g->step = 0;
    rng_seed_stream(g);
}
static void rng_pop_shadow(vro_rng* g)
{
    g->tag = g->saved_tag;
    g->step = g->saved_step;
    memcpy(g->s, g->saved_s, sizeof g->s);
    g->pair_valid = 0;
}
/* Improved marcher, march step n: word 0 = jitter (own engine in the reference), word 1 = roulette */
static double rng_march_jitter(vro_rng* g, uint32_t n)
{
    if (g->mode == VRO_RNG_REFERENCE) { double d; vro_minstd_canonical_double(g->jitter, &d); return d; }
    g->cur = n;
    return (double)unit24(rng_block(g, 0u, n)[0]);
}
static float rng_march_roulette(vro_rng* g)
{
    if (g->mode == VRO_RNG_REFERENCE) return vro_reference_get_float(g->mt, 0, 1);
    return (unit24(rng_block(g, 0u, g->cur)[1]) * (1.0f - 0.0f)) + 0.0f;
}

/* ------------------------------------------------------------------------------------ */
/* Utils (REF Utils/Utils.cpp)                                                          */
/* ------------------------------------------------------------------------------------ */

typedef struct v3 { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_mulv(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float r) { return v3_make(a.x * r, a.y * r, a.z * r); }

/* REF Utils.cpp:13-27 */
static inline v3 u_normalize(v3 v)
{
    float mag2 = v.x * v.x + v.y * v.y + v.z * v.z;
    if (mag2 > 0) {
        float invMag = 1 / sqrtf(mag2);
        return v3_make(v.x * invMag, v.y * invMag, v.z * invMag);
    }
    return v;
}
/* REF Utils.cpp:4-11 */
static inline v3 u_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* REF Utils.cpp:70-73 */
static inline float u_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* REF Utils.cpp:75-78: std::max(lo, std::min(hi, v)) */
static inline float u_clamp(float lo, float hi, float v)
{
    float m = (v < hi) ? v : hi;
    return (lo < m) ? m : lo;
}
/* REF Utils.cpp:80-83 */
static inline float u_deg2rad(float deg) { return (float)(deg * M_PI / 180); }

void vro_normalize(const float v[3], float out[3])
{
    v3 r = u_normalize(v3_make(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float vro_clamp(float lo, float hi, float v) { return u_clamp(lo, hi, v); }

/* REF Utils.cpp:29-62.  Camera set-up only (once per sample in the reference, constant per
 * frame): always libm. */
void vro_rotate(float pitch, float roll, float yaw, float p[3])
{
    pitch = (float)((pitch * M_PI) / 180);
    roll = (float)((roll * M_PI) / 180);
    yaw = (float)((yaw * M_PI) / 180);

    /* Utils.cpp sees only <cmath>'s std:: overloads, so under g++ (the build the golden
     * vectors come from) the unqualified cos/sin calls are the C double functions. */
    float cosa = (float)cos((double)yaw), sina = (float)sin((double)yaw);
    float cosb = (float)cos((double)pitch), sinb = (float)sin((double)pitch);
    float cosc = (float)cos((double)roll), sinc = (float)sin((double)roll);

    float Axx = cosa * cosb;
    float Axy = cosa * sinb * sinc - sina * cosc;
    float Axz = cosa * sinb * cosc + sina * sinc;
    float Ayx = sina * cosb;
    float Ayy = sina * sinb * sinc + cosa * cosc;
    float Ayz = sina * sinb * cosc - cosa * sinc;
    float Azx = -sinb;
    float Azy = cosb * sinc;
    float Azz = cosb * cosc;

    float px = p[0], py = p[1], pz = p[2];
    p[0] = Axx * px + Axy * py + Axz * pz;
    p[1] = Ayx * px + Ayy * py + Ayz * pz;
    p[2] = Azx * px + Azy * py + Azz * pz;
}

/* ------------------------------------------------------------------------------------ */
/* camera (REF integrators/BaseIntegrator.cpp:33-129, 287-312)                           */
/* ------------------------------------------------------------------------------------ */

/* REF BaseIntegrator.cpp:112-129 */
void vro_assign_point_to_quadrant(int i, int total, float out[3])
{
    out[0] = out[1] = out[2] = 0.0f;
    if (total < 4) return;
    if (total == 4) {
        if (i == 0) { out[0] = 0.25f;  out[1] = 0.25f; }
        if (i == 1) { out[0] = 0.25f;  out[1] = -0.25f; }
        if (i == 2) { out[0] = -0.25f; out[1] = 0.25f; }
        if (i == 3) { out[0] = -0.25f; out[1] = -0.25f; }
        return;
    }
    float rec[3];
    vro_assign_point_to_quadrant(i % (total / 4), total / 4, rec);
    float value = 1.0f / total;
    float sx, sy;
    if (i < total / 4.0f)          { sx = value;  sy = value; }
    else if (i < 2 * total / 4.0f) { sx = value;  sy = -value; }
    else if (i < 3 * total / 4.0f) { sx = -value; sy = value; }
    else                           { sx = -value; sy = -value; }
    out[0] = sx + rec[0];
    out[1] = sy + rec[1];
    out[2] = 0.0f + rec[2];
}

/* REF BaseIntegrator.cpp:288-312 */
void vro_camera_constants(const vr_options_pod* o, float out[4])
{
    float width = o->widthReference > 0.0f ? o->widthReference : o->width;
    float height = o->heightReference > 0.0f ? o->heightReference : o->height;
    float half_fov = (float)(o->fov * 0.5);
    float scale = tanf(u_deg2rad(half_fov));
    float imageAspectRatio = width / height;
    float x = (float)((2 * (0.5) / width - 1) * imageAspectRatio * scale);
    float y = (float)((1 - 2 * (0.5) / height) * scale);
    float xPlusOne = (float)((2 * (1.5) / width - 1) * imageAspectRatio * scale);
    float pixelWidth = xPlusOne - x;
    float yPlusOne = (float)((1 - 2 * (1.5) / height) * scale);
    float pixelHeight = yPlusOne - y;
    out[0] = scale; out[1] = imageAspectRatio; out[2] = pixelWidth; out[3] = pixelHeight;
}

/* REF BaseIntegrator.cpp:35-57 */
void vro_primary_direction(const vr_options_pod* o, int i, int j, uint32_t s, float dir[3])
{
    float cam[4];
    vro_camera_constants(o, cam);
    uint32_t width = o->widthReference > 0 ? o->widthReference : o->width;
    uint32_t height = o->heightReference > 0 ? o->heightReference : o->height;
    float x = (float)((2 * (i + 0.5) / (float)width - 1) * cam[1] * cam[0]);
    float y = (float)((1 - 2 * (j + 0.5) / (float)height) * cam[0]);
    v3 dn = u_normalize(v3_make(x, y, -1));
    float off[3];
    vro_assign_point_to_quadrant((int)s, (int)o->rayPerPixelCount, off);
    dir[0] = dn.x + off[0] * cam[2];
    dir[1] = dn.y + off[1] * cam[3];
    dir[2] = dn.z + 0.0f;
    if (o->cameraRotation[1] != 0 || o->cameraRotation[0] != 0 || o->cameraRotation[2] != 0)
        vro_rotate(o->cameraRotation[1], o->cameraRotation[0], o->cameraRotation[2], dir);
}

/* ------------------------------------------------------------------------------------ */
/* NanoVDB pieces (REF nanovdb/util/Ray.h, nanovdb/NanoVDB.h)                            */
/* ------------------------------------------------------------------------------------ */

typedef struct iray {
    v3    eye, dir, inv;
    float t0, t1;
} iray;

/* REF Ray.h:63-73 */
static inline iray ray_make(v3 eye, v3 dir, float t0, float t1)
{
    iray r;
    r.eye = eye; r.dir = dir;
    r.inv = v3_make(1 / dir.x, 1 / dir.y, 1 / dir.z);
    r.t0 = t0; r.t1 = t1;
    return r;
}
/* REF Ray.h:153-162 */
static inline v3 ray_at(const iray* r, float t)
{
    return v3_make(fmaf(t, r->dir.x, r->eye.x), fmaf(t, r->dir.y, r->eye.y), fmaf(t, r->dir.z, r->eye.z));
}
/* REF NanoVDB.h:1531-1536 */
static inline v3 mat_mult(const float* m, v3 p)
{
    return v3_make(fmaf(p.x, m[0], fmaf(p.y, m[1], p.z * m[2])),
                   fmaf(p.x, m[3], fmaf(p.y, m[4], p.z * m[5])),
                   fmaf(p.x, m[6], fmaf(p.y, m[7], p.z * m[8])));
}
/* REF NanoVDB.h:1547-1553 matMult(mat, vec, xyz) = Map::applyMapF (:2249) = Grid::indexToWorldF (:2641) */
static inline v3 index_to_world(const vr_grid_desc* g, v3 p)
{
    const float* m = g->mat;
    return v3_make(fmaf(p.x, m[0], fmaf(p.y, m[1], fmaf(p.z, m[2], g->vec[0]))),
                   fmaf(p.x, m[3], fmaf(p.y, m[4], fmaf(p.z, m[5], g->vec[1]))),
                   fmaf(p.x, m[6], fmaf(p.y, m[7], fmaf(p.z, m[8], g->vec[2]))));
}
void vro_index_to_world(const vr_grid_desc* g, const float* p, int n, float* out)
{
    for (int i = 0; i < n; ++i) {
        v3 w = index_to_world(g, v3_make(p[3 * i], p[3 * i + 1], p[3 * i + 2]));
        out[3 * i] = w.x; out[3 * i + 1] = w.y; out[3 * i + 2] = w.z;
    }
}
/* REF Ray.h:257-267 with NanoVDB.h:2262-2270 (applyInverseMapF / applyInverseJacobianF) */
static inline iray ray_world_to_index(const vr_grid_desc* g, v3 weye, v3 wdir)
{
    const float wt0 = 1e-5f;   /* Delta<float>, REF NanoVDB.h:706-709 */
    const float wt1 = FLT_MAX; /* Maximum<float> on the host, REF NanoVDB.h:744-747 */
    v3 eye = mat_mult(g->inv_mat, v3_make(weye.x - g->vec[0], weye.y - g->vec[1], weye.z - g->vec[2]));
    v3 dir = mat_mult(g->inv_mat, wdir);
    float length = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
    float invLength = 1.0f / length;
    float t1 = wt1;
    if (wt1 < FLT_MAX) t1 *= length;
    return ray_make(eye, v3_make(invLength * dir.x, invLength * dir.y, invLength * dir.z), length * wt0, t1);
}
/* REF Ray.h:397-423 (CoordBBox overload: max padded by one voxel) and :512-520 */
static inline int ray_clip(iray* r, const vr_grid_desc* g)
{
    float t0 = r->t0, t1 = r->t1;
    const float eye[3] = { r->eye.x, r->eye.y, r->eye.z };
    const float inv[3] = { r->inv.x, r->inv.y, r->inv.z };
    for (int i = 0; i < 3; ++i) {
        float a = (float)g->bbox_min[i], b = (float)(g->bbox_max[i] + 1);
        if (a >= b) return 0;
        a = (a - eye[i]) * inv[i];
        b = (b - eye[i]) * inv[i];
        if (a > b) { float tmp = a; a = b; b = tmp; }
        if (a > t0) t0 = a;
        if (b < t1) t1 = b;
        if (t0 > t1) return 0;
    }
    r->t0 = t0; r->t1 = t1;
    return 1;
}
/* REF NanoVDB.h:818-821,1133 (Coord::Floor) + :5248-5263 (ReadAccessor::getValue): nearest
 * voxel by floor, background outside the tree.  Dense restatement: outside bbox -> background. */
static inline float grid_value(const vr_grid_desc* g, v3 p)
{
    int32_t i = (int32_t)floorf(p.x), j = (int32_t)floorf(p.y), k = (int32_t)floorf(p.z);
    if (i < g->bbox_min[0] || i > g->bbox_max[0] || j < g->bbox_min[1] || j > g->bbox_max[1] ||
        k < g->bbox_min[2] || k > g->bbox_max[2])
        return g->background;
    int64_t ny = (int64_t)g->bbox_max[1] - g->bbox_min[1] + 1;
    int64_t nz = (int64_t)g->bbox_max[2] - g->bbox_min[2] + 1;
    int64_t idx = ((int64_t)(i - g->bbox_min[0]) * ny + (j - g->bbox_min[1])) * nz + (k - g->bbox_min[2]);
    return g->voxels[idx];
}
/* REF nanovdb/util/GridStats.h:765-838 (getExtrema(...).max()) for a dense box */
float vro_grid_max(const vr_grid_desc* g)
{
    if (g->max_value >= 0.0f) return g->max_value;
    int64_t n = ((int64_t)g->bbox_max[0] - g->bbox_min[0] + 1) * ((int64_t)g->bbox_max[1] - g->bbox_min[1] + 1) *
                ((int64_t)g->bbox_max[2] - g->bbox_min[2] + 1);
    float m = g->voxels[0];
    for (int64_t i = 1; i < n; ++i) if (g->voxels[i] > m) m = g->voxels[i];
    return m;
}

static void iray_export(const iray* r, float out[8])
{
    out[0] = r->eye.x; out[1] = r->eye.y; out[2] = r->eye.z;
    out[3] = r->dir.x; out[4] = r->dir.y; out[5] = r->dir.z;
    out[6] = r->t0; out[7] = r->t1;
}
int vro_world_ray_to_index_clip(const vr_grid_desc* g, const float eye[3], const float dir[3], float out[8])
{
    iray r = ray_world_to_index(g, v3_make(eye[0], eye[1], eye[2]), v3_make(dir[0], dir[1], dir[2]));
    int hit = ray_clip(&r, g);
    iray_export(&r, out);
    return hit;
}
int vro_index_ray_clip(const vr_grid_desc* g, const float eye[3], const float dir[3], float out[8])
{
    iray r = ray_make(v3_make(eye[0], eye[1], eye[2]), v3_make(dir[0], dir[1], dir[2]), 1e-5f, FLT_MAX);
    int hit = ray_clip(&r, g);
    iray_export(&r, out);
    return hit;
}
float vro_grid_value_at(const vr_grid_desc* g, const float p[3])
{
    return grid_value(g, v3_make(p[0], p[1], p[2]));
}

/* ------------------------------------------------------------------------------------ */
/* phase function and direction sampling                                                */
/* ------------------------------------------------------------------------------------ */

/* REF Utils/PhaseFunction.cpp:4-8 */
static inline float hg_phase(float g, float cos_theta)
{
    float denom = 1 + g * g - 2 * g * cos_theta;
    return (float)(1 / (4 * M_PI) * (1 - g * g) / (denom * sqrtf(denom)));
}
float vro_henyey_greenstein(float g, float cos_theta) { return hg_phase(g, cos_theta); }

/* REF Utils/DirectionSampler.cpp:38-72 with Utils/ONB.cpp:5-22; xi1, xi2 ~ U[0, 0.9999) */
static v3 hg_sample(float g, v3 direction, float xi, float xi2)
{
    /* ONB::Update(direction): basis from the UN-normalised direction */
    v3 n = direction, s;
    if (fabsf(n.x) > fabsf(n.z)) { s.x = -n.y; s.y = n.x; s.z = 0; }
    else                         { s.x = 0; s.y = -n.z; s.z = n.y; }
    s = u_normalize(s);
    v3 t = u_cross(n, s);

    v3 normalizedDirection = u_normalize(direction);
    float cos_theta;
    if (g != 0.0f) {
        float aux = ((1 - g * g) / (1 - g + 2 * g * xi));
        cos_theta = (1 + g * g - (aux * aux)) / (2 * g);
    } else {
        cos_theta = (float)(1.0 - 2.0 * xi);
    }
    float c2 = 1.0f - cos_theta * cos_theta;
    float sin_theta = sqrtf((0.0f < c2) ? c2 : 0.0f);
    float phi = (float)((xi2 * 2) * M_PI);
    v3 local;
    if (g_math_mode == VRO_MATH_LIBM) {
        /* under g++ `sin_theta * cos(phi)` is float * double (see vro_rotate), narrowed by Vec3f's ctor */
        local = v3_make((float)(sin_theta * cos((double)phi)), (float)(sin_theta * sin((double)phi)), cos_theta);
    } else {
        local = v3_make(sin_theta * vro_shared_cosf(phi), sin_theta * vro_shared_sinf(phi), cos_theta);
    }
    v3 v1 = v3_scale(s, local.x);
    v3 v2 = v3_scale(t, local.y);
    v3 v3_ = v3_scale(normalizedDirection, local.z);
    return v3_add(v3_add(v1, v2), v3_);
}
void vro_sample_henyey_greenstein(float g, const float dir[3], float xi1, float xi2, float out[3])
{
    v3 r = hg_sample(g, v3_make(dir[0], dir[1], dir[2]), xi1, xi2);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ------------------------------------------------------------------------------------ */
/* Perlin noise (REF Utils/PerlinNoiseSampler.cpp)                                       */
/* ------------------------------------------------------------------------------------ */

/* REF PerlinNoiseSampler.cpp:11-28 — Ken Perlin's published reference permutation (data) */
static const uint8_t perlin_perm[256] = {
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225,
    140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148,
    247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32,
    57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175,
    74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122,
    60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54,
    65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18, 169,
    200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64,
    52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212,
    207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213,
    119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9,
    129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104,
    218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241,
    81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157,
    184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93,
    222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180
};
static inline int pp(int i) { return perlin_perm[i & 255]; } /* p[256+i] = p[i], REF :82-83 */

/* REF PerlinNoiseSampler.cpp:32-40 */
static inline float pn_fade(float t) { return t * t * t * (t * (t * 6 - 15) + 10); }
static inline float pn_lerp(float t, float a, float b) { return a + t * (b - a); }
static inline float pn_grad(int hash, float x, float y, float z)
{
    int h = hash & 15;
    float u = h < 8 ? x : y;
    float v = h < 4 ? y : (h == 12 || h == 14) ? x : z;
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}
/* REF PerlinNoiseSampler.cpp:46-68 */
float vro_perlin_noise(float x, float y, float z)
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
/* REF PerlinNoiseSampler.cpp:94-98 */
float vro_perlin_eval_density(const float p[3])
{
    float freq = 1.8f;
    return (float)((1 + vro_perlin_noise(p[0] * freq, p[1] * freq, p[2] * freq)) * 0.5);
}

/* The synthetic cloud of BASELINE configs 2-4 (not in the reference):
 *   rho(p) = smoothstep(1 - |p|/R) * max(0, fbm(p)),
 *   fbm(p) = sum_{o=0..3} 0.5^o * noise(p * f0 * 2^o),  f0 = 4/n  (4 noise cells across the grid),
 * p = voxel centre relative to the grid centre, smoothstep(t) = t*t*(3-2t) on clamp(t,0,1)
 * evaluated in float.  The GPU generator (csrc/vr_gen.hip) must reproduce it bit for bit. */
static inline float cloud_density(int n, float R, int i, int j, int k)
{
    float half = 0.5f * (float)n;
    float px = ((float)i + 0.5f) - half, py = ((float)j + 0.5f) - half, pz = ((float)k + 0.5f) - half;
    float r = sqrtf(px * px + py * py + pz * pz);
    float t = 1.0f - r / R;
    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    float fall = t * t * (3.0f - 2.0f * t);
    if (!(fall > 0.0f)) return 0.0f;
    float f = 4.0f / (float)n, amp = 1.0f, sum = 0.0f;
    for (int o = 0; o < 4; ++o) {
        sum = sum + amp * vro_perlin_noise(px * f + 17.0f, py * f + 17.0f, pz * f + 17.0f);
        f = f * 2.0f;
        amp = amp * 0.5f;
    }
    float d = fall * (sum > 0.0f ? sum : 0.0f);
    return d;
}
typedef struct cloud_job { int n; float R; float* vox; int i0, i1; } cloud_job;
static void* cloud_worker(void* arg)
{
    cloud_job* jb = (cloud_job*)arg;
    int n = jb->n;
    for (int i = jb->i0; i < jb->i1; ++i)
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k)
                jb->vox[((int64_t)i * n + j) * n + k] = cloud_density(n, jb->R, i, j, k);
    return NULL;
}
void vro_make_cloud(int n, float radius_voxels, float* voxels, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    pthread_t th[64];
    cloud_job jobs[64];
    for (int t = 0; t < nthreads; ++t) {
        jobs[t].n = n; jobs[t].R = radius_voxels; jobs[t].vox = voxels;
        jobs[t].i0 = (int)((int64_t)n * t / nthreads);
        jobs[t].i1 = (int)((int64_t)n * (t + 1) / nthreads);
        pthread_create(&th[t], NULL, cloud_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------------------------------------ */
/* closest hit (stand-in for rtcIntersect1, REF Utils/EmbreeHelper.cpp:4-36;             */
/* Embree 4.0.1 is a prebuilt Windows binary absent from the checkout: PARITY UNPINNED)  */
/* ------------------------------------------------------------------------------------ */

/* Moeller-Trumbore, both faces, hit when 0 <= t < current best (tnear = 0, tfar = inf,
 * REF EmbreeHelper.cpp:13).  Ties keep the lower primitive id.  The GPU BVH uses the same
 * per-triangle arithmetic, so the closest t is identical. */
static inline int tri_hit(const float* v0, const float* v1, const float* v2, const float org[3],
                          const float dir[3], float* t_out)
{
    float e1x = v1[0] - v0[0], e1y = v1[1] - v0[1], e1z = v1[2] - v0[2];
    float e2x = v2[0] - v0[0], e2y = v2[1] - v0[1], e2z = v2[2] - v0[2];
    float px = dir[1] * e2z - dir[2] * e2y;
    float py = dir[2] * e2x - dir[0] * e2z;
    float pz = dir[0] * e2y - dir[1] * e2x;
    float det = e1x * px + e1y * py + e1z * pz;
    if (det == 0.0f) return 0;
    float inv = 1.0f / det;
    float tx = org[0] - v0[0], ty = org[1] - v0[1], tz = org[2] - v0[2];
    float u = (tx * px + ty * py + tz * pz) * inv;
    if (u < 0.0f || u > 1.0f) return 0;
    float qx = ty * e1z - tz * e1y;
    float qy = tz * e1x - tx * e1z;
    float qz = tx * e1y - ty * e1x;
    float v = (dir[0] * qx + dir[1] * qy + dir[2] * qz) * inv;
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = (e2x * qx + e2y * qy + e2z * qz) * inv;
    if (!(t >= 0.0f)) return 0;
    *t_out = t;
    return 1;
}
int vro_intersect_closest(const vr_mesh_desc* m, const float org[3], const float dir[3], float* t_out)
{
    int best = -1;
    float tbest = INFINITY;
    for (uint32_t p = 0; p < m->ntris; ++p) {
        const uint32_t* idx = m->tri_idx + 3 * (size_t)p;
        float t;
        if (tri_hit(m->verts_xyz + 3 * (size_t)idx[0], m->verts_xyz + 3 * (size_t)idx[1],
                    m->verts_xyz + 3 * (size_t)idx[2], org, dir, &t) && t < tbest) {
            tbest = t; best = (int)p;
        }
    }
    if (best >= 0) *t_out = tbest;
    return best;
}

/* ------------------------------------------------------------------------------------ */
/* per-path state (REF integrators/BaseIntegrator.h:21-50, HandleIntersectionData)       */
/* ------------------------------------------------------------------------------------ */

typedef struct path_ctx {
    const vro_scene*      sc;
    const vr_options_pod* o;
    vro_rng*              rng;
    vr_render_stats*      st;
    v3    rayOrigin, rayDirection;
    int   objectId, previousObjectId;
    v3    hitPoint;
    float tFar;
    float rayPDF;
    float sigma_max;  /* densityExtrema.max() */
} path_ctx;

static inline v3 opt3(const float* p) { return v3_make(p[0], p[1], p[2]); }

/* ------------------------------------------------------------------------------------ */
/* mesh-bounded media for the trackers — NO REFERENCE COUNTERPART                        */
/* ------------------------------------------------------------------------------------ */
/* The reference's trackers clip to the grid bbox only (REF IntegratorDeltaTracking.cpp:37-39,
 * IntegratorRatioTracking.cpp:37-39); SURVEY 8(d) config 5 defines the capability: the walk's
 * segment is the bbox interval cut by the mesh's entry/exit interval.  Entry and exit are found
 * by the walk the reference's MARCHERS do before they march (REF
 * IntegratorHomogeneousRayMarcherNEE.cpp:16-60, 104-108): closest hit; a hit on another object
 * than the last one is an entry and moves the origin to hit + dir * 0.001; the next hit on the
 * same object is the exit.  A ray that starts inside a medium (scattered rays, shadow rays)
 * starts with that medium's object as "the last one", so its first hit on it is the exit.
 * Opaque shapes (dissolve <= 0) are black absorbers (the marchers return black there).
 * Returns the interval [*s0, *s1] in units of |dir| from `org`. */
enum { SEG_NONE = 0, SEG_FOUND = 1, SEG_BLACK = 2 };
static int mesh_segment(path_ctx* d, v3 org, v3 dir, int inside_id, float* s0, float* s1, int* medium)
{
    const vr_mesh_desc* m = d->sc->mesh;
    int object = inside_id;
    float s_acc = 0.0f, start = 0.0f;
    for (int guard = 0; guard < 64; ++guard) {
        float o3[3] = { org.x, org.y, org.z }, d3[3] = { dir.x, dir.y, dir.z }, t;
        d->st->ray_queries++;
        int prim = vro_intersect_closest(m, o3, d3, &t);                    /* REF EmbreeHelper.cpp:18 */
        if (prim < 0) return SEG_NONE;
        int previous = object;                                              /* REF :24 */
        object = (int)m->prim_to_shape[prim];                               /* REF :27 */
        if (!(m->shape_dissolve[object] > 0)) return SEG_BLACK;             /* REF NEE.cpp:52, :118 */
        if (previous == object) {                                           /* REF NEE.cpp:55 exit hit */
            *s0 = start; *s1 = s_acc + t; *medium = object;
            return SEG_FOUND;
        }
        v3 hit = v3_add(v3_scale(dir, t), org);                             /* REF EmbreeHelper.cpp:28 */
        org = v3_add(hit, v3_scale(dir, 0.001f));                           /* REF NEE.cpp:105 */
        s_acc = (s_acc + t) + 0.001f;
        start = s_acc;
    }
    return SEG_NONE;
}
/* cut the index-space ray by the mesh interval of the matching world-space ray; scale = index
 * parameter per world parameter.  Returns SEG_*; SEG_FOUND with an empty interval sets *empty. */
static int mesh_clip(path_ctx* d, iray* r, v3 worg, v3 wdir, float scale, int inside_id, int* medium, int* empty)
{
    float s0 = 0, s1 = 0;
    *empty = 0;
    int rc = mesh_segment(d, worg, wdir, inside_id, &s0, &s1, medium);
    if (rc != SEG_FOUND) return rc;
    float a = s0 * scale, b = s1 * scale;
    if (a > r->t0) r->t0 = a;
    if (b < r->t1) r->t1 = b;
    if (r->t0 > r->t1) *empty = 1;
    return rc;
}

/* Safety cap (not in the reference, which would spin forever): a path whose t stops
 * advancing is cut and counted in stats->step_limit_hits. */
#define VRO_STEP_LIMIT 200000000ull

/* ------------------------------------------------------------------------------------ */
/* delta tracking (REF integrators/IntegratorDeltaTracking.cpp:11-176)                   */
/* ------------------------------------------------------------------------------------ */
/* STUDY SWITCHES (default 0 = the reference's tracker as written; tests/test_local_majorant_oracle.py).  Each turns ONE
 * thing in which the reference's delta tracker differs from a textbook one into the textbook form, so that the
 * difference between the reference's image and the local-majorant tracker's (cast_ray_delta_local below) can be
 * measured instead of attributed:
 *   VRO_DELTA_LOOKUP_AFTER_STEP  the collision is classified with the density AT the tentative collision, not with the
 *                                one looked up one free path earlier (REF :55 reads sigma at tFar, :74 advances tFar,
 *                                :84-98 classify with the old sigma)
 *   VRO_DELTA_CLAMP_ENTRY        a lookup on the bbox's max face belongs to the last voxel: Ray::clip pads the box to
 *                                max + 1 (Ray.h:401), so a ray that enters through a max face starts at index max + 1,
 *                                where getValue is the background, and crosses its first 10 * stepSizeMin in an
 *                                empty-space step (REF :70-72) */
static int g_delta_variant = 0;
void vro_set_delta_variant(int bits) { g_delta_variant = bits; }
int  vro_get_delta_variant(void) { return g_delta_variant; }

static inline float grid_value_clamped(const vr_grid_desc* g, v3 p)
{
    int32_t c[3] = { (int32_t)floorf(p.x), (int32_t)floorf(p.y), (int32_t)floorf(p.z) };
    for (int a = 0; a < 3; ++a) { if (c[a] < g->bbox_min[a]) c[a] = g->bbox_min[a]; if (c[a] > g->bbox_max[a]) c[a] = g->bbox_max[a]; }
    return grid_value(g, v3_make((float)c[0], (float)c[1], (float)c[2]));
}

static v3 cast_ray_delta(path_ctx* d, int mesh_bound)
{
    const vr_options_pod* o = d->o;
    const vr_grid_desc* grid = d->sc->density;
    float densityMultiplier = o->lightRayDensityMultiplier;                 /* :13 */
    d->rayPDF = 1.0f;                                                       /* :14 */
    int hasEmission = d->sc->temperature != NULL;                           /* :15 */
    int depthRemaining = (int)o->maxDepth;                                  /* :17 */
    float tMin = o->stepSizeMin, tMax = o->stepSizeMax;                     /* :19-20 */
    v3 rayDirection = u_normalize(d->rayDirection);                         /* :22 */

    iray iRay = ray_world_to_index(grid, d->rayOrigin, d->rayDirection);    /* :30-35 */
    if (!ray_clip(&iRay, grid)) return opt3(o->backgroundColor);           /* :37-39 */
    int medium = -1;
    if (mesh_bound) {   /* no reference counterpart: see mesh_segment */
        v3 idir = mat_mult(grid->inv_mat, d->rayDirection);
        float length = sqrtf(idir.x * idir.x + idir.y * idir.y + idir.z * idir.z);   /* as Ray.h:262-266 scales t0 */
        int empty, rc = mesh_clip(d, &iRay, d->rayOrigin, d->rayDirection, length, -1, &medium, &empty);
        if (rc == SEG_BLACK) return v3_make(0, 0, 0);
        if (rc == SEG_NONE || empty) return opt3(o->backgroundColor);
    }

    v3 result = v3_make(0, 0, 0);                                           /* :41 */
    float sigma_maj = d->sigma_max * (o->sigma_a + o->sigma_s);             /* :44-47 */
    float tFar = iRay.t0;                                                   /* :50 */
    int terminated = 0;
    uint64_t steps = 0;

    /* VRO_RNG_PHILOX_GEOM (the FAST kernel's GEOM variant, vr_kernels.h): when EVERY free path is clamped to stepSizeMin
     * (-ln(0.00001) is the largest draw of :63) the steps are independent Bernoulli trials with P(real collision) = 1 - pNull,
     * and the walk draws ONE exponential clock instead of two numbers per step: a step costs -ln(pNull), the real collision
     * is the step at which the clock runs out, and a second number of the same draw decides absorption / scattering.  Same
     * distribution of every step's outcome as the reference's loop below, another stream.  Restated step by step here. */
    const int geom = d->rng->mode == VRO_RNG_PHILOX_GEOM && !o->useImportanceSampling &&
                     sigma_maj > 0.0f && 11.52f * o->stepSizeMultiplier / sigma_maj <= tMin;
    double geom_clock = 0.0;
    float  geom_v = 0.0f;
    int    geom_fresh = 1;            /* a new walk: draw its clock */

    while (!terminated && depthRemaining > 0) {                             /* :54 */
        if (++steps > VRO_STEP_LIMIT) { d->st->step_limit_hits++; terminated = 1; break; }
        rng_begin_step(d->rng);
        if (geom && geom_fresh) {
            uint32_t w[2];
            vro_xoshiro128p_next2(d->rng->s, w);
            geom_clock = -log((double)((unit24(w[0]) * (1.0f - 0.00001f)) + 0.00001f));
            geom_v = unit24(w[1]);
            geom_fresh = 0;
        }
        const int clamp_entry = (g_delta_variant & VRO_DELTA_CLAMP_ENTRY) != 0;
        float sigma = densityMultiplier * (clamp_entry ? grid_value_clamped(grid, ray_at(&iRay, tFar))
                                                       : grid_value(grid, ray_at(&iRay, tFar))); /* :55 */
        d->st->density_lookups++;
        float mu_a = sigma * o->sigma_a;
        float mu_s = sigma * o->sigma_s;
        float mu_t = mu_a + mu_s;

        float pathLength = 0;
        if (sigma > 0.0f) {                                                 /* :62-69 */
            if (geom) pathLength = tMin;                                    /* every draw is clamped to it */
            else {
                pathLength = -m_logf(rng_free_path(d->rng)) / sigma_maj;
                pathLength *= o->stepSizeMultiplier;
                pathLength = u_clamp(tMin, tMax, pathLength);
            }
        } else {
            pathLength = tMin * 10;                                         /* :71 */
        }
        tFar += pathLength;                                                 /* :74 */
        if (tFar > iRay.t1) break;                                          /* :77-79 */
        if (sigma <= 0.0f) continue;                                        /* :81-82 */
        if (g_delta_variant & VRO_DELTA_LOOKUP_AFTER_STEP) {                /* study switch, not the reference */
            sigma = densityMultiplier * (clamp_entry ? grid_value_clamped(grid, ray_at(&iRay, tFar)) : grid_value(grid, ray_at(&iRay, tFar)));
            d->st->density_lookups++;
            mu_a = sigma * o->sigma_a; mu_s = sigma * o->sigma_s; mu_t = mu_a + mu_s;
        }

        float pAbsorption = mu_a / sigma_maj;                               /* :84-86 */
        float pScattering = mu_s / sigma_maj;
        float pn = 1 - pAbsorption - pScattering;
        float pNull = (0 < pn) ? pn : 0;

        double sampleAttenuation = m_expf(-(pathLength)*mu_t);              /* :88 */
        d->rayPDF = (float)(d->rayPDF * (mu_t * sampleAttenuation));        /* :91 */

        float sample;
        if (geom) {
            /* the clock: this step costs -ln(pNull); it is a real collision when the clock runs out, and then
             * `sample` is uniform in [pNull, 1) */
            geom_clock -= -log((double)pNull);
            if (geom_clock >= 0.0) sample = -1.0f;                          /* null collision */
            else { sample = pNull + geom_v * (1.0f - pNull); if (!(sample >= pNull)) sample = pNull; }
        } else
            sample = rng_event(d->rng);                                     /* :93 */
        if (sample < pNull) {
            /* null collision */
        } else if (sample < pNull + pAbsorption) {                          /* :101-113 */
            if (hasEmission) {
                (void)grid_value(d->sc->temperature, ray_at(&iRay, tFar)); /* :103 (value unused) */
                d->st->density_lookups++;
                result = v3_add(result, v3_scale(opt3(o->emissionColor), mu_a)); /* :105-107 */
            }
            result = v3_add(result, opt3(o->mediumColor));                  /* :110 */
            terminated = 1;
        } else {                                                            /* :115-158 */
            if (depthRemaining-- < 0) { terminated = 1; break; }            /* :118 */
            d->st->scatter_events++;
            float xi1, xi2;
            rng_direction(d->rng, &xi1, &xi2);
            v3 scattered = hg_sample(o->heyneyGreensteinG, d->rayDirection, xi1, xi2); /* :124-127 */
            v3 iRayOrigin = ray_at(&iRay, tFar);                            /* :130 */
            if (o->useImportanceSampling) {                                 /* :137-144 */
                float cos_theta = u_dot(rayDirection, d->rayDirection);
                d->rayPDF *= hg_phase(o->heyneyGreensteinG, cos_theta);
            }
            d->rayDirection = scattered;                                    /* :146 */
            iRay = ray_make(iRayOrigin, scattered, 1e-5f, FLT_MAX);         /* :147 */
            if (!ray_clip(&iRay, grid)) { terminated = 1; break; }          /* :150-155 */
            if (mesh_bound) {
                int empty, rc = mesh_clip(d, &iRay, index_to_world(grid, iRayOrigin), mat_mult(grid->mat, scattered), 1.0f,
                                          medium, &medium, &empty);
                if (rc == SEG_BLACK || empty) { terminated = 1; break; }
            }
            tFar = iRay.t0;                                                 /* :157 */
            geom_fresh = 1;
        }
    }
    if (terminated) return result;                                          /* :170-173 */
    return v3_add(result, opt3(o->backgroundColor));                        /* :175 */
}

/* ------------------------------------------------------------------------------------ */
/* delta tracking over per-cell majorants — NO REFERENCE COUNTERPART                      */
/* ------------------------------------------------------------------------------------ */
/* Scalar restatement of the estimator behind VR_MODE_FAST_LOCAL / VR_MODE_EXACT_LOCAL (north_star "DensityFields":
 * a coarse majorant grid walked by a DDA; volume_renderer_amd/csrc/vr_kernels.h, vr_delta_dda_kernel).  What it replaces
 * in the reference is the ONE global majorant of REF IntegratorDeltaTracking.cpp:44-47; everything around the free-path
 * sampling follows the reference's delta tracker (prologue :13-52, absorption :101-113, scattering :115-158, the
 * scatter budget :54 / :175).  The estimator, in the order the kernel executes it, one generator stream per PATH:
 *   - the grid is cut into cells of (1 << cell_log)^3 voxels from bbox.min; a cell's majorant is the maximum of its
 *     voxels (padding to whole 8^3 bricks counts as background) times lightRayDensityMultiplier * (sigma_a + sigma_s);
 *   - the ray walks the cells with a 3-D DDA (tm* = parameter of the cell's far face per axis, td* = per-cell increment);
 *   - in a cell with majorant mu > 0 one transition of the stream gives two words: free path t' = t - ln(u) / mu from
 *     the first, the collision type from the second; a free path that ends beyond min(far face, t1) is dropped and the
 *     walk goes on from the face (the exponential is memoryless) — or ends as escaped at t1; an empty cell is one step;
 *   - at a tentative collision the density is looked up AT the collision (voxel index clamped into the current cell, so
 *     that rounding on a face can never pair a voxel with a smaller neighbour's majorant): absorbed if xi * mu < mu_a,
 *     scattered if xi * mu < mu_t, else null;
 *   - a scatter takes the next transition of the same stream for sampleHenyeyGreenstein.
 * VRO_MATH_SHARED + VRO_RNG_PHILOX is bit-identical to the kernel's EXACT build (tests/test_gpu_local_majorant.py). */
static struct {
    const vr_grid_desc* grid;
    int    cell_log, nc[3];
    float* major;
} g_local;

/* the backend's rule (vr_backend.hip, render_row_set): cells of 64^3 voxels on grids of 1024 and more, 32^3 from 512, 16^3 from 256, else the 8^3 bricks */
int vro_local_auto_cell_log(const vr_grid_desc* g)
{
    int maxdim = 0;
    for (int a = 0; a < 3; ++a) { int d = g->bbox_max[a] - g->bbox_min[a] + 1; if (d > maxdim) maxdim = d; }
    return 3 + (maxdim >= 1024 ? 3 : (maxdim >= 512 ? 2 : (maxdim >= 256 ? 1 : 0)));
}
void vro_local_majorants_free(void)
{
    free(g_local.major);
    memset(&g_local, 0, sizeof g_local);
}
/* per-cell maxima of `g` for cells of (1 << cell_log)^3 voxels; cell_log in [3, 7] */
int vro_local_majorants_build(const vr_grid_desc* g, int cell_log)
{
    if (!g || !g->voxels || cell_log < 3 || cell_log > 7) return VR_ERR_INVALID_ARG;
    vro_local_majorants_free();
    int dim[3], pad[3];
    for (int a = 0; a < 3; ++a) {
        dim[a] = g->bbox_max[a] - g->bbox_min[a] + 1;
        pad[a] = (dim[a] + 7) / 8 * 8;                                   /* the backend stores whole 8^3 bricks */
        g_local.nc[a] = (pad[a] / 8 + (1 << (cell_log - 3)) - 1) >> (cell_log - 3);
    }
    size_t n = (size_t)g_local.nc[0] * g_local.nc[1] * g_local.nc[2];
    g_local.major = (float*)malloc(n * sizeof(float));
    if (!g_local.major) return VR_ERR_OUT_OF_MEMORY;
    for (size_t c = 0; c < n; ++c) g_local.major[c] = -INFINITY;
    for (int i = 0; i < pad[0]; ++i)
        for (int j = 0; j < pad[1]; ++j)
            for (int k = 0; k < pad[2]; ++k) {
                float v = (i < dim[0] && j < dim[1] && k < dim[2]) ? g->voxels[((size_t)i * dim[1] + j) * dim[2] + k] : g->background;
                float* m = &g_local.major[((size_t)(i >> cell_log) * g_local.nc[1] + (j >> cell_log)) * g_local.nc[2] + (k >> cell_log)];
                *m = fmaxf(*m, v);
            }
    g_local.grid = g; g_local.cell_log = cell_log;
    return VR_OK;
}

typedef struct local_dda {
    float tmx, tmy, tmz, tdx, tdy, tdz;
    int   bi, bj, bk;
    float mu;
} local_dda;

/* kernel: set_ray + enter_brick */
static void local_dda_start(local_dda* s, const vr_grid_desc* grid, const iray* r, float t, float mu_scale)
{
    const int   cl = g_local.cell_log;
    const float cs = (float)(1 << cl);
    const v3 q = ray_at(r, t);
    int c[3] = { ((int32_t)floorf(q.x) - grid->bbox_min[0]) >> cl, ((int32_t)floorf(q.y) - grid->bbox_min[1]) >> cl,
                 ((int32_t)floorf(q.z) - grid->bbox_min[2]) >> cl };
    for (int a = 0; a < 3; ++a) c[a] = c[a] < 0 ? 0 : (c[a] >= g_local.nc[a] ? g_local.nc[a] - 1 : c[a]);   /* the first point lies ON the bbox */
    s->bi = c[0]; s->bj = c[1]; s->bk = c[2];
    const v3 e = r->eye, dd = r->dir;
    const float ix = dd.x != 0 ? 1 / dd.x : INFINITY, iy = dd.y != 0 ? 1 / dd.y : INFINITY, iz = dd.z != 0 ? 1 / dd.z : INFINITY;
    s->tdx = cs * fabsf(ix); s->tdy = cs * fabsf(iy); s->tdz = cs * fabsf(iz);
    s->tmx = dd.x != 0 ? ((float)(grid->bbox_min[0] + ((c[0] + (dd.x > 0 ? 1 : 0)) << cl)) - e.x) * ix : INFINITY;
    s->tmy = dd.y != 0 ? ((float)(grid->bbox_min[1] + ((c[1] + (dd.y > 0 ? 1 : 0)) << cl)) - e.y) * iy : INFINITY;
    s->tmz = dd.z != 0 ? ((float)(grid->bbox_min[2] + ((c[2] + (dd.z > 0 ? 1 : 0)) << cl)) - e.z) * iz : INFINITY;
    s->mu = g_local.major[((size_t)c[0] * g_local.nc[1] + c[1]) * g_local.nc[2] + c[2]] * mu_scale;
}
/* density at p, voxel index clamped into cell (bi, bj, bk) and into the bbox (a cell may stick out of it; the ray's interval does not) */
static float local_cell_value(const vr_grid_desc* g, v3 p, const local_dda* s)
{
    const int cl = g_local.cell_log;
    const int cell[3] = { s->bi, s->bj, s->bk };
    int32_t v[3] = { (int32_t)floorf(p.x) - g->bbox_min[0], (int32_t)floorf(p.y) - g->bbox_min[1], (int32_t)floorf(p.z) - g->bbox_min[2] };
    for (int a = 0; a < 3; ++a) {
        const int32_t lo = cell[a] << cl, last = g->bbox_max[a] - g->bbox_min[a];
        int32_t hi = lo + (1 << cl) - 1;
        if (hi > last) hi = last;
        v[a] = v[a] > hi ? hi : v[a];
        v[a] = v[a] < lo ? lo : v[a];
    }
    int64_t ny = (int64_t)g->bbox_max[1] - g->bbox_min[1] + 1, nz = (int64_t)g->bbox_max[2] - g->bbox_min[2] + 1;
    return g->voxels[((int64_t)v[0] * ny + v[1]) * nz + v[2]];
}

/* mesh_bound (VRO_INTEGRATOR_MESH_DELTA_LOCAL, the estimator of meshDeltaTracking in the LOCAL modes): the walk's segments are cut
 * to the mesh's entry/exit interval exactly where cast_ray_delta(d, 1) cuts them — no reference counterpart either way. */
static v3 cast_ray_delta_local(path_ctx* d, int mesh_bound)
{
    const vr_options_pod* o = d->o;
    const vr_grid_desc* grid = d->sc->density;
    vro_rng* g = d->rng;
    const float sig_t = o->sigma_a + o->sigma_s;
    const float mu_scale = o->lightRayDensityMultiplier * sig_t;
    const int   hasEmission = d->sc->temperature != NULL;                   /* REF :15 */
    int depthRemaining = (int)o->maxDepth;                                  /* REF :17 */

    iray iRay = ray_world_to_index(grid, d->rayOrigin, d->rayDirection);    /* REF :30-35 */
    if (!ray_clip(&iRay, grid)) return opt3(o->backgroundColor);           /* REF :37-39 */
    int medium = -1;
    if (mesh_bound) {
        v3 idir = mat_mult(grid->inv_mat, d->rayDirection);
        float length = sqrtf(idir.x * idir.x + idir.y * idir.y + idir.z * idir.z);
        int empty, rc = mesh_clip(d, &iRay, d->rayOrigin, d->rayDirection, length, -1, &medium, &empty);
        if (rc == SEG_BLACK) return v3_make(0, 0, 0);
        if (rc == SEG_NONE || empty) return opt3(o->backgroundColor);
    }
    if (o->maxDepth == 0) return opt3(o->backgroundColor);                  /* REF :54 */

    float t = iRay.t0;                                                      /* REF :50 */
    local_dda s;
    local_dda_start(&s, grid, &iRay, t, mu_scale);
    uint64_t steps = 0;
    for (;;) {
        if (++steps > VRO_STEP_LIMIT) { d->st->step_limit_hits++; return v3_make(0, 0, 0); }
        const float tface = fminf(fminf(s.tmx, s.tmy), s.tmz);
        const float te = fminf(tface, iRay.t1);
        float t_next = INFINITY, xi = 0.0f;
        if (s.mu > 0.0f) {
            float u;
            if (g->mode == VRO_RNG_REFERENCE) {
                u = vro_reference_get_float(g->mt, 0.00001f, 1.0f);         /* getFloat(0.00001, 1), REF :63 */
                xi = vro_reference_get_float(g->mt, 0.0f, 1.0f);            /* getFloat(0, 1), REF :93 */
            } else {
                uint32_t w[2];
                vro_xoshiro128p_next2(g->s, w);
                u = (unit24(w[0]) * (1.0f - 0.00001f)) + 0.00001f;
                xi = unit24(w[1]);
            }
            t_next = t + (-m_logf(u)) / s.mu;
        }
        if (!(t_next < te)) {
            /* no collision inside this cell: on to the next one */
            t = te;
            if (tface >= iRay.t1) return opt3(o->backgroundColor);          /* REF :77-79, :175 the ray has left the grid */
            if (s.tmx <= s.tmy && s.tmx <= s.tmz) { s.bi += iRay.dir.x > 0 ? 1 : -1; s.tmx += s.tdx; }
            else if (s.tmy <= s.tmz)              { s.bj += iRay.dir.y > 0 ? 1 : -1; s.tmy += s.tdy; }
            else                                  { s.bk += iRay.dir.z > 0 ? 1 : -1; s.tmz += s.tdz; }
            if (s.bi < 0 || s.bi >= g_local.nc[0] || s.bj < 0 || s.bj >= g_local.nc[1] || s.bk < 0 || s.bk >= g_local.nc[2])
                return opt3(o->backgroundColor);                            /* (rounding: the last face and t1 coincide) */
            s.mu = g_local.major[((size_t)s.bi * g_local.nc[1] + s.bj) * g_local.nc[2] + s.bk] * mu_scale;
            continue;
        }
        /* tentative collision at t_next: real with probability mu_t(x) / mu */
        t = t_next;
        d->st->density_lookups++;
        const float sigma = o->lightRayDensityMultiplier * local_cell_value(grid, ray_at(&iRay, t), &s);
        const float thr = xi * s.mu;
        const float mu_a = sigma * o->sigma_a;
        if (thr < mu_a) {                                                   /* REF :101-113 */
            v3 colour = opt3(o->mediumColor);
            if (hasEmission) colour = v3_add(v3_scale(opt3(o->emissionColor), mu_a), colour);
            return colour;
        }
        if (!(thr < sigma * sig_t)) continue;                               /* null collision */
        /* scattering                                                          REF :115-158 */
        d->st->scatter_events++;
        depthRemaining--;
        float xi1, xi2;
        if (g->mode == VRO_RNG_REFERENCE) {
            xi1 = vro_reference_get_float(g->mt, 0.0f, 0.9999f);
            xi2 = vro_reference_get_float(g->mt, 0.0f, 0.9999f);
        } else {
            uint32_t w[2];
            vro_xoshiro128p_next2(g->s, w);
            xi1 = (unit24(w[0]) * (0.9999f - 0.0f)) + 0.0f;
            xi2 = (unit24(w[1]) * (0.9999f - 0.0f)) + 0.0f;
        }
        v3 scattered = hg_sample(o->heyneyGreensteinG, d->rayDirection, xi1, xi2);   /* REF :124-127 */
        v3 point = ray_at(&iRay, t);                                        /* REF :130 */
        d->rayDirection = scattered;                                        /* REF :146 */
        iRay = ray_make(point, scattered, 1e-5f, FLT_MAX);                  /* REF :147 */
        if (!ray_clip(&iRay, grid)) return v3_make(0, 0, 0);                /* REF :150-155 "scattering failed" */
        if (mesh_bound) {
            int empty, rc = mesh_clip(d, &iRay, index_to_world(grid, point), mat_mult(grid->mat, scattered), 1.0f, medium, &medium, &empty);
            if (rc == SEG_BLACK || empty) return v3_make(0, 0, 0);
        }
        if (depthRemaining <= 0) return opt3(o->backgroundColor);           /* REF :54, :175 */
        t = iRay.t0;                                                        /* REF :157 */
        local_dda_start(&s, grid, &iRay, t, mu_scale);
    }
}

/* ------------------------------------------------------------------------------------ */
/* ratio tracking (REF integrators/IntegratorRatioTracking.cpp)                          */
/* ------------------------------------------------------------------------------------ */

/* REF IntegratorRatioTracking.cpp:213-283 (maxStepSize and sigma_maj are unused there) */
static float direct_lightning_ray_march(path_ctx* d, const iray* iRay, int mesh_bound, int medium)
{
    const vr_options_pod* o = d->o;
    const vr_grid_desc* grid = d->sc->density;
    float transmission = 1.0f;
    float densityMultiplier = o->shadowRayDensityMultiplier;                /* :215 */
    float tMin = o->stepSizeMin * 5;                                        /* :216 */
    float tMax = o->stepSizeMax;
    d->st->shadow_rays++;

    v3 lightDirection = opt3(o->lightPosition);                             /* :221-225 */
    iray lightRay = ray_make(ray_at(iRay, iRay->t0), lightDirection, 1e-5f, FLT_MAX); /* :226 */
    if (!ray_clip(&lightRay, grid)) return transmission;                   /* :229-232 */
    if (mesh_bound) {
        int empty, m2 = medium;
        int rc = mesh_clip(d, &lightRay, index_to_world(grid, lightRay.eye), mat_mult(grid->mat, lightDirection), 1.0f,
                           medium, &m2, &empty);
        if (rc == SEG_BLACK) return 0.0f;          /* an opaque surface between the point and the light */
        if (empty) return transmission;
    }
    float tFar = lightRay.t0;                                               /* :234 */
    uint64_t steps = 0;

    for (;;) {                                                              /* :236 */
        if (++steps > VRO_STEP_LIMIT) { d->st->step_limit_hits++; return transmission; }
        rng_begin_step(d->rng);
        float sigma = densityMultiplier * grid_value(grid, ray_at(&lightRay, tFar)); /* :238 */
        d->st->density_lookups++;
        float mu_a = sigma * o->sigma_a;
        float mu_s = sigma * o->sigma_s;
        float mu_t = mu_a + mu_s;
        float stepSize = 0;
        if (sigma > 0.0f) {                                                 /* :244-250 */
            stepSize = -m_logf(rng_free_path(d->rng)) / mu_t;
            stepSize *= o->stepSizeMultiplier;
            stepSize = u_clamp(tMin, tMax, stepSize);
        } else {
            stepSize = tMin * 100;                                          /* :252 */
        }
        tFar += stepSize;                                                   /* :255 */
        if (tFar > lightRay.t1) return transmission;                        /* :258-260 */
        if (sigma <= 0.0f) continue;                                        /* :262-263 */
        float transmissionStep = m_expf(-(stepSize)*mu_t);                  /* :266 */
        transmission *= transmissionStep;                                   /* :269 */
        if (transmission < 0.05f) {                                         /* :272-279 */
            float q = 0.75f;
            if (rng_event(d->rng) < q) transmission = 0.0f;
            else transmission /= 1.0f - q;
        }
        if (transmission <= 0.0f) return transmission;                      /* :281-282 */
    }
}

/* REF IntegratorRatioTracking.cpp:11-211 */
static v3 cast_ray_ratio(path_ctx* d, int mesh_bound)
{
    const vr_options_pod* o = d->o;
    const vr_grid_desc* grid = d->sc->density;
    float densityMultiplier = o->lightRayDensityMultiplier;                 /* :13 */
    d->rayPDF = 1.0f;
    int hasEmission = d->sc->temperature != NULL;                           /* :15 */
    int depthRemaining = (int)o->maxDepth;                                  /* :17 */
    float tMin = o->stepSizeMin, tMax = o->stepSizeMax;                     /* :19-20 */

    iray iRay = ray_world_to_index(grid, d->rayOrigin, d->rayDirection);    /* :30-35 */
    if (!ray_clip(&iRay, grid)) return opt3(o->backgroundColor);           /* :37-39 */
    int medium = -1;
    if (mesh_bound) {   /* no reference counterpart: see mesh_segment */
        v3 idir = mat_mult(grid->inv_mat, d->rayDirection);
        float length = sqrtf(idir.x * idir.x + idir.y * idir.y + idir.z * idir.z);
        int empty, rc = mesh_clip(d, &iRay, d->rayOrigin, d->rayDirection, length, -1, &medium, &empty);
        if (rc == SEG_BLACK) return v3_make(0, 0, 0);
        if (rc == SEG_NONE || empty) return opt3(o->backgroundColor);
    }

    float sigma_maj = d->sigma_max * (o->sigma_a + o->sigma_s);             /* :42-45 */
    float tFar = iRay.t0;                                                   /* :48 */
    v3 radiance = v3_make(0, 0, 0);                                         /* :50 */
    float transmission = 1.0f;                                              /* :51 */
    int terminated = 0;
    uint64_t steps = 0;

    while (!terminated && depthRemaining > 0) {                             /* :55 */
        if (++steps > VRO_STEP_LIMIT) { d->st->step_limit_hits++; terminated = 1; break; }
        rng_begin_step(d->rng);
        if (transmission < 0.05f) {                                         /* :57-64 */
            float q = 0.75f;
            if (rng_roulette(d->rng) < q) transmission = 0.0f;
            else transmission /= 1.0f - q;
        }
        if (transmission <= 0.0f) { terminated = 1; break; }                /* :66-69 */

        float sigma = densityMultiplier * grid_value(grid, ray_at(&iRay, tFar)); /* :71 */
        d->st->density_lookups++;
        float mu_a = sigma * o->sigma_a;
        float mu_s = sigma * o->sigma_s;
        float mu_t = mu_a + mu_s;

        float pathLength = 0;
        if (sigma > 0.0f) {                                                 /* :77-83 */
            pathLength = -m_logf(rng_free_path(d->rng)) / sigma_maj;
            pathLength *= o->stepSizeMultiplier;
            pathLength = u_clamp(tMin, tMax, pathLength);
        } else {
            pathLength = tMin * 10;                                         /* :85 */
        }
        tFar += pathLength;                                                 /* :88 */
        if (tFar > iRay.t1) break;                                          /* :91-93 */
        if (sigma <= 0.0f) continue;                                        /* :95-96 */

        float pAbsorption = mu_a / sigma_maj;                               /* :98-100 */
        float pScattering = mu_s / sigma_maj;
        float pn = 1 - pAbsorption - pScattering;
        float pNull = (0 < pn) ? pn : 0;

        double sampleAttenuation = m_expf(-(pathLength)*mu_t);              /* :103 */
        transmission = (float)(transmission * sampleAttenuation);          /* :105 */
        d->rayPDF = (float)(d->rayPDF * (mu_t * sampleAttenuation));        /* :106 */

        float sample = rng_event(d->rng);                                   /* :108 */

        if (hasEmission) {                                                  /* :110-122 */
            float emission = grid_value(d->sc->temperature, ray_at(&iRay, tFar));
            d->st->density_lookups++;
            if (emission > 0.0f) {
                v3 e = v3_scale(opt3(o->emissionColor), transmission * mu_a);
                e = v3_scale(e, emission);
                e = v3_scale(e, pathLength);
                radiance = v3_add(radiance, e);
            }
        }

        if (sample < pNull) {
            /* null collision (:125) */
        } else if (sample < pNull + pAbsorption) {                          /* :127-131 */
            radiance = v3_add(radiance, v3_scale(opt3(o->mediumColor), transmission));
            terminated = 1;
        } else {                                                            /* :133-191 */
            if (depthRemaining-- < 0) { terminated = 1; break; }            /* :136 */
            d->st->scatter_events++;
            if (transmission > 0.0f) {                                      /* :141 */
                rng_push_shadow(d->rng);
                float lightTransmission = direct_lightning_ray_march(d, &iRay, mesh_bound, medium); /* :143 */
                rng_pop_shadow(d->rng);
                float cos_theta = u_dot(d->rayDirection, opt3(o->lightPosition)); /* :144 */
                float hgPDF = hg_phase(o->heyneyGreensteinG, cos_theta);   /* :145 */
                v3 nee = v3_scale(opt3(o->lightColor), transmission * lightTransmission); /* :147-152 */
                nee = v3_scale(nee, pathLength);
                nee = v3_scale(nee, hgPDF);
                radiance = v3_add(radiance, nee);

                float xi1, xi2;
                rng_direction(d->rng, &xi1, &xi2);
                v3 scattered = hg_sample(o->heyneyGreensteinG, d->rayDirection, xi1, xi2); /* :156-159 */
                v3 iRayOrigin = ray_at(&iRay, tFar);                        /* :162 */
                if (o->useImportanceSampling) {                             /* :169-176 */
                    float c = u_dot(scattered, d->rayDirection);
                    d->rayPDF *= hg_phase(o->heyneyGreensteinG, c);
                }
                d->rayDirection = scattered;                                /* :178 */
                iRay = ray_make(iRayOrigin, scattered, 1e-5f, FLT_MAX);     /* :179 */
                if (!ray_clip(&iRay, grid)) { terminated = 1; break; }      /* :182-187 */
                if (mesh_bound) {
                    int empty, rc = mesh_clip(d, &iRay, index_to_world(grid, iRayOrigin), mat_mult(grid->mat, scattered), 1.0f,
                                              medium, &medium, &empty);
                    if (rc == SEG_BLACK || empty) { terminated = 1; break; }
                }
                tFar = iRay.t0;                                             /* :189 */
            }
        }
    }
    if (terminated) return radiance;                                        /* :194-195 */
    return v3_add(radiance, v3_scale(opt3(o->backgroundColor), transmission)); /* :208 */
}

/* ------------------------------------------------------------------------------------ */
/* fixed-step marchers through a mesh-bounded homogeneous medium                         */
/* (REF IntegratorHomogeneousRayMarcherNEE.cpp, IntegratorHomogeneousRayMarcherImproved.cpp) */
/* ------------------------------------------------------------------------------------ */

/* REF Utils/EmbreeHelper.cpp:4-36 */
static int cast_single_ray(path_ctx* d)
{
    float org[3] = { d->rayOrigin.x, d->rayOrigin.y, d->rayOrigin.z };
    float dir[3] = { d->rayDirection.x, d->rayDirection.y, d->rayDirection.z };
    float t;
    d->st->ray_queries++;
    int prim = vro_intersect_closest(d->sc->mesh, org, dir, &t);
    if (prim < 0) return 0;
    d->previousObjectId = d->objectId;                                      /* :24 */
    d->objectId = (int)d->sc->mesh->prim_to_shape[prim];                    /* :27 */
    d->hitPoint = v3_add(v3_scale(d->rayDirection, t), d->rayOrigin);       /* :28 */
    d->tFar = t;                                                            /* :32 */
    return 1;
}

/* The march between entry and exit hit.
 * improved = 0: REF IntegratorHomogeneousRayMarcherNEE.cpp:57-102
 * improved = 1: REF IntegratorHomogeneousRayMarcherImproved.cpp:58-133 */
static v3 march_segment(path_ctx* d, int improved)
{
    const vr_options_pod* o = d->o;
    float transparency = 1;
    v3 result = v3_make(0, 0, 0);
    float distance = d->tFar;                                               /* NEE :59 */
    v3 light_dir = opt3(o->lightPosition);
    v3 light_color = opt3(o->lightColor);
    float sigma_a = o->sigma_a, sigma_s = o->sigma_s;
    float density = 1.0f;                                                   /* Improved :69 */
    float g = o->heyneyGreensteinG;

    float step_size = 0.2f;                                                 /* NEE :66-68 */
    int ns = (int)ceilf(distance / step_size);
    step_size = distance / ns;

    v3 rayDirection = u_normalize(d->rayDirection);                         /* NEE :70-71 */
    v3 rayOrigin = d->rayOrigin;

    for (size_t n = 0; n < (size_t)ns; n++) {                               /* NEE :73 */
        float t;
        if (improved) {
            double randomOffset = rng_march_jitter(d->rng, (uint32_t)n);      /* Improved :82-84 */
            t = (float)(step_size * (n + randomOffset));                    /* Improved :90 */
        } else {
            t = step_size * (n + 0.5f);                                     /* NEE :75 */
        }
        v3 samplePosition = v3_add(rayOrigin, v3_scale(rayDirection, t));   /* NEE :76 */
        float sampleAttenuation = improved ? m_expf(-step_size * density * (sigma_a + sigma_s)) /* Improved :95 */
                                           : m_expf(-step_size * sigma_a);  /* NEE :79 */
        transparency *= sampleAttenuation;                                  /* NEE :82 */

        d->rayDirection = light_dir;                                        /* NEE :84 */
        /* `samplePosition + data->rayDirection * 0.001`: Vec3f * double-literal narrows to float */
        d->rayOrigin = v3_add(samplePosition, v3_scale(d->rayDirection, 0.001f)); /* NEE :85 */
        d->st->shadow_rays++;
        if (cast_single_ray(d)) {                                           /* NEE :88 */
            float dl = d->tFar;
            if (improved) {
                float cos_theta = u_dot(rayDirection, light_dir);           /* Improved :107 */
                float light_attenuation = m_expf(-dl * density * (sigma_a + sigma_s)); /* :108 */
                v3 c = v3_scale(light_color, light_attenuation);            /* :110-117 */
                c = v3_scale(c, density);
                c = v3_scale(c, sigma_s);
                c = v3_scale(c, hg_phase(g, cos_theta));
                c = v3_scale(c, transparency);
                c = v3_scale(c, step_size);
                result = v3_add(result, c);
            } else {
                float light_attenuation = m_expf(-dl * sigma_a);            /* NEE :91 */
                v3 c = v3_scale(light_color, transparency);                 /* NEE :93-97 */
                c = v3_scale(c, light_attenuation);
                c = v3_scale(c, step_size);
                result = v3_add(result, c);
            }
        }
        if (improved) {                                                     /* Improved :121-129 */
            int dd = 2;
            if (transparency < 1e-3) {
                if (rng_march_roulette(d->rng) > 1.f / dd) n = (size_t)ns;
                else transparency *= dd;
            }
        }
    }
    return v3_add(v3_scale(opt3(o->backgroundColor), transparency), result); /* NEE :102 */
}

/* The march of IntegratorHeterogeneousPerlinNoise::handleIntersection (REF
 * integrators/IntegratorHeterogeneousPerlinNoise.cpp:62-150): the Improved marcher's walk with the
 * density read from PerlinNoiseSampler::eval_density at every sample and along every light ray.
 * Unlike the homogeneous marchers it does NOT rescale the step to distance/ns. */
static v3 march_segment_perlin(path_ctx* d)
{
    const vr_options_pod* o = d->o;
    float distance = d->tFar;                                               /* :64 */
    v3 light_dir = opt3(o->lightPosition);
    v3 light_color = opt3(o->lightColor);
    float sigma_s = o->sigma_s;
    float sigma_t = o->sigma_a + o->sigma_s;                                /* :74 */
    float g = o->heyneyGreensteinG;
    float step_size = 0.2f;                                                 /* :78 */
    int ns = (int)ceilf(distance / step_size);                              /* :79 */
    float transparency = 1;
    v3 result = v3_make(0, 0, 0);
    v3 rayDirection = u_normalize(d->rayDirection);                         /* :86 */
    v3 rayOrigin = d->rayOrigin;

    for (size_t n = 0; n < (size_t)ns; n++) {                               /* :89 */
        double randomOffset = rng_march_jitter(d->rng, (uint32_t)n);        /* :91-93 */
        float t = (float)(step_size * (n + randomOffset));                  /* :99 */
        v3 samplePosition = v3_add(rayOrigin, v3_scale(rayDirection, t));   /* :101 */
        float sp[3] = { samplePosition.x, samplePosition.y, samplePosition.z };
        float density = vro_perlin_eval_density(sp);                        /* :103 */
        d->st->density_lookups++;
        float sampleAttenuation = m_expf(-step_size * density * sigma_t);   /* :106 */
        transparency *= sampleAttenuation;                                  /* :109 */
        d->rayDirection = light_dir;                                        /* :112 */
        d->rayOrigin = v3_add(samplePosition, v3_scale(d->rayDirection, 0.001f)); /* :113 */
        if (density > 0 && (d->st->shadow_rays++, cast_single_ray(d))) {    /* :116-117 */
            float tau = 0;
            float dl = d->tFar;                                             /* :119 */
            float num_steps_light = ceilf(dl / step_size);                  /* :120 */
            for (size_t nl = 0; (float)nl < num_steps_light; ++nl) {        /* :122 */
                float tLight = (float)(step_size * (nl + 0.5));             /* :123 */
                v3 pl = v3_add(samplePosition, v3_scale(light_dir, tLight)); /* :124 (float * Vec3f) */
                float lp[3] = { pl.x, pl.y, pl.z };
                tau += vro_perlin_eval_density(lp);                         /* :125 */
                d->st->density_lookups++;
            }
            float cos_theta = u_dot(rayDirection, light_dir);               /* :128 */
            float light_attenuation = m_expf(-tau * step_size * sigma_t);   /* :129 */
            v3 c = v3_scale(light_color, light_attenuation);                /* :130-137, left to right */
            c = v3_scale(c, density);
            c = v3_scale(c, sigma_s);
            c = v3_scale(c, hg_phase(g, cos_theta));
            c = v3_scale(c, transparency);
            c = v3_scale(c, step_size);
            result = v3_add(result, c);
            int dd = 2;                                                     /* :140-148 */
            if (transparency < 1e-3) {
                if (rng_march_roulette(d->rng) > 1.f / dd) n = (size_t)ns;
                else transparency *= dd;
            }
        }
    }
    return v3_add(v3_scale(opt3(o->backgroundColor), transparency), result); /* :153 */
}

/* IntegratorDensitySampling (REF integrators/IntegratorDensitySampling.cpp:7-78): analytic Beer-Lambert
 * between entry and exit hit of ANY object (no dissolve test), tinted by the material's diffuse colour. */
static v3 cast_ray_density_sampling(path_ctx* d)
{
    const vr_options_pod* o = d->o;
    const vr_mesh_desc* m = d->sc->mesh;
    for (int guard = 0; guard < 4096; ++guard) {
        if (0 >= o->maxDepth) return v3_make(0, 0, 0);                      /* :9-11 (depth stays 0) */
        if (!cast_single_ray(d)) return opt3(o->backgroundColor);          /* :13-42 */
        if (d->previousObjectId == d->objectId) {                           /* :50 exit intersection */
            float distance = d->tFar;
            float sigma_a = o->sigma_a;
            const float* df = m->shape_diffuse ? m->shape_diffuse + 3 * (size_t)d->objectId : NULL;
            v3 scatteringColor = df ? v3_make(df[0], df[1], df[2]) : v3_make(0, 0, 0); /* :54 */
            float transmission = m_expf(-distance * sigma_a);               /* :55 */
            return v3_add(v3_scale(opt3(o->backgroundColor), transmission), v3_scale(scatteringColor, 1 - transmission)); /* :57 */
        }
        d->rayOrigin = v3_add(d->hitPoint, v3_scale(d->rayDirection, 0.001f)); /* :62 */
    }
    return v3_make(0, 0, 0);
}

/* castRay + handleIntersection unrolled into a loop
 * (REF IntegratorHomogeneousRayMarcherNEE.cpp:10-118; Improved :11-149 is the same walk). */
static v3 cast_ray_marcher(path_ctx* d, int improved)
{
    const vr_options_pod* o = d->o;
    const vr_mesh_desc* m = d->sc->mesh;
    uint32_t depth = 0;
    for (int guard = 0; guard < 4096; ++guard) {
        if (depth >= o->maxDepth) return v3_make(0, 0, 0);                  /* :12-14 */
        if (!cast_single_ray(d)) return opt3(o->backgroundColor);          /* :25-44 */
        float dissolve = m->shape_dissolve[d->objectId];                    /* :49 */
        if (dissolve > 0) {                                                 /* :52 */
            if (d->previousObjectId == d->objectId)                         /* :55 */
                return improved == 2 ? march_segment_perlin(d) : march_segment(d, improved);
            d->rayOrigin = v3_add(d->hitPoint, v3_scale(d->rayDirection, 0.001f)); /* :105 */
            continue;                                                       /* :107 */
        }
        if (d->previousObjectId == d->objectId) {                           /* :112-117 */
            d->rayOrigin = v3_add(d->rayOrigin, v3_scale(d->rayDirection, 0.001f));
            depth += 1;
            continue;
        }
        /* REF :118 — the reference falls off the end of a non-void function here (opaque
         * surface, first hit): undefined.  The restatement and the GPU path return black. */
        return v3_make(0, 0, 0);
    }
    return v3_make(0, 0, 0);
}

/* ------------------------------------------------------------------------------------ */
/* fixed-step marchers through the voxel grid                                            */
/* (REF integrators/IntegratorNanoVDBSimple.cpp:10-95, IntegratorNanoVDBEmission.cpp:10-110; */
/*  the two differ in the density scale and the emission term)                           */
/* ------------------------------------------------------------------------------------ */
static v3 cast_ray_grid_marcher(path_ctx* d, int emission_variant)
{
    const vr_options_pod* o = d->o;
    const vr_grid_desc* grid = d->sc->density;
    const vr_grid_desc* tgrid = d->sc->temperature;
    v3 rayDirection = u_normalize(d->rayDirection);                         /* Simple :12 */
    iray iRay = ray_world_to_index(grid, d->rayOrigin, d->rayDirection);    /* :15-20 */
    if (!ray_clip(&iRay, grid)) return opt3(o->backgroundColor);           /* :22-24 */
    float density = emission_variant ? 64.0f : 0.5f;                        /* Simple :25, Emission :35 */
    float lightRayDensity = density * 0.5f;                                 /* :26 */
    const float step_size = 0.5f;                                           /* :28 */
    float transmittance = 1.0f;                                             /* :30 */
    v3 result = v3_make(0, 0, 0);                                           /* :32 */
    v3 light = opt3(o->lightPosition);
    uint32_t n = 0;                                                         /* march step index (Philox counter) */

    for (float t = iRay.t0; t < iRay.t1; t += step_size, ++n) {             /* :34 */
        v3 ip = ray_at(&iRay, t);
        float sigma = grid_value(grid, ip) * density;                       /* :36 */
        d->st->density_lookups++;
        float emissionValue = 0.0f;
        v3 emission = v3_make(0, 0, 0);
        if (emission_variant) {                                             /* Emission :47-48 */
            emissionValue = grid_value(tgrid, ip) * 4.0f;
            d->st->density_lookups++;
            emission = v3_scale(opt3(o->emissionColor), emissionValue);
        }
        float sampleAttenuation = m_expf(-step_size * sigma);               /* :38 */
        transmittance *= sampleAttenuation;                                 /* :40 */
        v3 world = index_to_world(grid, ip);                                /* :42 */
        d->rayDirection = light;                                            /* :44 */
        d->rayOrigin = world;                                               /* :45 */

        if (sigma > 0 || (emission_variant && emissionValue > 0)) {         /* Simple :47, Emission :61 */
            /* castLightRay (:97-128): the light ray in index space, clipped to the bbox */
            iray lRay = ray_world_to_index(grid, d->rayOrigin, d->rayDirection);
            d->st->shadow_rays++;
            if (ray_clip(&lRay, grid)) {                                    /* :49 */
                float tau = 0;
                float light_step_size = step_size * 20.0f;                  /* :52 */
                float num_steps_light = ceilf(lRay.t1 / light_step_size);   /* :53 (t1, not t1 - t0) */
                for (size_t nl = 0; (float)nl < num_steps_light; ++nl) {    /* :55 */
                    float tLight = (float)(light_step_size * (nl + 0.5));   /* :56 (double product) */
                    tau += grid_value(grid, ray_at(&lRay, lRay.t0 + tLight)) * lightRayDensity; /* :57 */
                    d->st->density_lookups++;
                }
                float cos_theta = u_dot(rayDirection, light);               /* :60 */
                float light_attenuation = m_expf(-tau * light_step_size * sigma); /* :61 */
                v3 c = v3_scale(opt3(o->lightColor), light_attenuation);    /* :63-69, left to right */
                c = v3_scale(c, sigma);
                c = v3_scale(c, hg_phase(o->heyneyGreensteinG, cos_theta));
                c = v3_scale(c, transmittance);
                c = v3_scale(c, step_size);
                result = v3_add(result, c);
            } else {                                                        /* :73-78 */
                v3 c = v3_scale(opt3(o->lightColor), sigma);
                c = v3_scale(c, transmittance);
                c = v3_scale(c, step_size);
                result = v3_add(result, c);
            }
            if (emission_variant)                                           /* Emission :90 */
                result = v3_add(result, v3_scale(v3_scale(emission, transmittance), step_size));
            int dd = 2;                                                     /* :81-89 */
            if (transmittance < 1e-3) {
                float u;
                if (d->rng->mode == VRO_RNG_REFERENCE) u = vro_reference_get_float(d->rng->mt, 0, 1);
                else u = (unit24(rng_block(d->rng, 0u, n)[1]) * (1.0f - 0.0f)) + 0.0f;
                if (u > 1.f / dd) t = iRay.t1;
                else transmittance *= dd;
            }
        }
    }
    return v3_add(result, v3_scale(opt3(o->backgroundColor), transmittance)); /* :94 */
}

/* ------------------------------------------------------------------------------------ */
/* render loop (REF integrators/BaseIntegrator.cpp)                                      */
/* ------------------------------------------------------------------------------------ */

static v3 cast_ray(path_ctx* d, int integrator)
{
    d->st->path_samples++;
    switch (integrator) {
    case VR_INTEGRATOR_HOMOGENEOUS_RAY_MARCHER_NEE:      return cast_ray_marcher(d, 0);
    case VR_INTEGRATOR_HOMOGENEOUS_RAY_MARCHER_IMPROVED: return cast_ray_marcher(d, 1);
    case VR_INTEGRATOR_RATIO_TRACKING:                   return cast_ray_ratio(d, 0);
    case VR_INTEGRATOR_MESH_RATIO_TRACKING:              return cast_ray_ratio(d, 1);
    case VR_INTEGRATOR_MESH_DELTA_TRACKING:              return cast_ray_delta(d, 1);
    case VR_INTEGRATOR_NANOVDB_SIMPLE:                   return cast_ray_grid_marcher(d, 0);
    case VR_INTEGRATOR_NANOVDB_EMISSION:                 return cast_ray_grid_marcher(d, 1);
    case VR_INTEGRATOR_HETEROGENEOUS_PERLIN_NOISE:       return cast_ray_marcher(d, 2); /* same castRay walk (:12-60, :155-168) */
    case VR_INTEGRATOR_DENSITY_SAMPLING:                 return cast_ray_density_sampling(d);
    case VRO_INTEGRATOR_DELTA_LOCAL:                     return cast_ray_delta_local(d, 0);
    case VRO_INTEGRATOR_MESH_DELTA_LOCAL:                return cast_ray_delta_local(d, 1);
    default:                                             return cast_ray_delta(d, 0);
    }
}

static int scene_ok(const vro_scene* sc, int integrator)
{
    if (integrator == VRO_INTEGRATOR_DELTA_LOCAL || integrator == VRO_INTEGRATOR_MESH_DELTA_LOCAL) {
        if (!sc->density) return VR_ERR_NO_DENSITY;
        if (integrator == VRO_INTEGRATOR_MESH_DELTA_LOCAL && !sc->mesh) return VR_ERR_NO_MESH;
        if (g_local.grid != sc->density || !g_local.major) return VR_ERR_INVALID_ARG;   /* vro_local_majorants_build first */
        return VR_OK;
    }
    if (integrator < 0 || integrator >= VR_INTEGRATOR_COUNT) return VR_ERR_INVALID_ARG;
    if (integrator == VR_INTEGRATOR_HETEROGENEOUS_PERLIN_NOISE || integrator == VR_INTEGRATOR_DENSITY_SAMPLING) { if (!sc->mesh) return VR_ERR_NO_MESH; }
    else if (integrator == VR_INTEGRATOR_MESH_RATIO_TRACKING || integrator == VR_INTEGRATOR_MESH_DELTA_TRACKING) {
        if (!sc->density) return VR_ERR_NO_DENSITY;
        if (!sc->mesh) return VR_ERR_NO_MESH;
    }
    else if (integrator >= VR_INTEGRATOR_RATIO_TRACKING) {
        if (!sc->density) return VR_ERR_NO_DENSITY;
        if (integrator == VR_INTEGRATOR_NANOVDB_EMISSION && !sc->temperature) return VR_ERR_NO_DENSITY; /* Emission :13-18 exits */
    }
    else if (!sc->mesh) return VR_ERR_NO_MESH;
    return VR_OK;
}

static uint32_t g_jitter_state = 1u; /* the function-static default_random_engine, seed 1 */

/* renderRay (REF BaseIntegrator.cpp:33-110) for one pixel */
static void render_pixel(path_ctx* d, int integrator, int i, int j, const float cam[4], float* pix,
                         float* per_sample)
{
    const vr_options_pod* o = d->o;
    uint32_t width = o->widthReference > 0 ? o->widthReference : o->width;     /* :35 */
    uint32_t height = o->heightReference > 0 ? o->heightReference : o->height; /* :36 */
    float x = (float)((2 * (i + 0.5) / (float)width - 1) * cam[1] * cam[0]);  /* :38 */
    float y = (float)((1 - 2 * (j + 0.5) / (float)height) * cam[0]);          /* :39 */
    uint32_t raysPerPixel = o->rayPerPixelCount;
    v3 dirNormalized = u_normalize(v3_make(x, y, -1));                         /* :43 */
    v3 color = v3_make(0, 0, 0);

    for (uint32_t s = 0; s < raysPerPixel; s++) {                              /* :49 */
        float off[3];
        vro_assign_point_to_quadrant((int)s, (int)raysPerPixel, off);          /* :51 */
        float dir[3] = { dirNormalized.x + off[0] * cam[2], dirNormalized.y + off[1] * cam[3],
                         dirNormalized.z + 0.0f };                             /* :53 */
        if (o->cameraRotation[1] != 0 || o->cameraRotation[0] != 0 || o->cameraRotation[2] != 0)
            vro_rotate(o->cameraRotation[1], o->cameraRotation[0], o->cameraRotation[2], dir); /* :56-57 */
        d->rayOrigin = opt3(o->cameraPosition);                                /* :61 */
        d->rayDirection = v3_make(dir[0], dir[1], dir[2]);                     /* :62 */
        d->objectId = -1;                                                      /* :63 */
        if (d->rng->mode != VRO_RNG_REFERENCE) rng_begin_path(d->rng, (uint32_t)i, (uint32_t)j, s);
        v3 r = cast_ray(d, integrator);
        if (o->useImportanceSampling) {                                        /* :66-69, :75-79 */
            float pdf = d->rayPDF;
            float den = (0.1f < pdf) ? pdf : 0.1f;
            r = v3_make(r.x / den, r.y / den, r.z / den);
        }
        if (per_sample) { per_sample[3 * s] = r.x; per_sample[3 * s + 1] = r.y; per_sample[3 * s + 2] = r.z; }
        color = v3_add(color, r);                                              /* :71 */
    }
    float n = (float)raysPerPixel;                                             /* :92 */
    pix[0] = color.x / n; pix[1] = color.y / n; pix[2] = color.z / n;          /* :108 */
}

int vro_cast_ray_sample(const vro_scene* sc, const vr_options_pod* o, int integrator, uint64_t seed,
                        int i, int j, uint32_t s, float rgb[3], vr_render_stats* stats)
{
    int rc = scene_ok(sc, integrator);
    if (rc) return rc;
    vr_render_stats local; memset(&local, 0, sizeof local);
    vro_rng rng; memset(&rng, 0, sizeof rng);
    rng.mode = VRO_RNG_PHILOX;
    rng.key[0] = (uint32_t)seed; rng.key[1] = (uint32_t)(seed >> 32);
    path_ctx d; memset(&d, 0, sizeof d);
    d.sc = sc; d.o = o; d.rng = &rng; d.st = stats ? stats : &local;
    d.sigma_max = sc->density ? vro_grid_max(sc->density) : 0.0f;
    float dir[3];
    vro_primary_direction(o, i, j, s, dir);
    d.rayOrigin = opt3(o->cameraPosition);
    d.rayDirection = v3_make(dir[0], dir[1], dir[2]);
    d.objectId = -1;
    rng_begin_path(&rng, (uint32_t)i, (uint32_t)j, s);
    v3 r = cast_ray(&d, integrator);
    if (o->useImportanceSampling) {
        float den = (0.1f < d.rayPDF) ? d.rayPDF : 0.1f;
        r = v3_make(r.x / den, r.y / den, r.z / den);
    }
    rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
    return VR_OK;
}

/* renderPartial (REF BaseIntegrator.cpp:287-325) */
int vro_render_rows(const vro_scene* sc, const vr_options_pod* o, int integrator, int rng_mode,
                    uint64_t seed, uint32_t from_row, uint32_t to_row, float* pix, float* per_sample,
                    vr_render_stats* stats)
{
    int rc = scene_ok(sc, integrator);
    if (rc) return rc;
    if (o->rayPerPixelCount == 0 || o->width == 0) return VR_ERR_INVALID_ARG;
    vr_render_stats local; memset(&local, 0, sizeof local);
    float cam[4];
    vro_camera_constants(o, cam);

    uint32_t mt[625];
    vro_rng rng; memset(&rng, 0, sizeof rng);
    rng.mode = rng_mode;
    rng.key[0] = (uint32_t)seed; rng.key[1] = (uint32_t)(seed >> 32);
    rng.mt = mt;
    rng.jitter = &g_jitter_state;
    if (rng_mode == VRO_RNG_REFERENCE) vro_mt19937_seed(mt, from_row);        /* :298 */

    path_ctx d; memset(&d, 0, sizeof d);
    d.sc = sc; d.o = o; d.rng = &rng; d.st = &local;
    d.rayPDF = 1.0f;                                                           /* :301 */
    d.sigma_max = sc->density ? vro_grid_max(sc->density) : 0.0f;

    uint32_t spp = o->rayPerPixelCount;
    for (uint32_t j = from_row; j < to_row; ++j) {                             /* :314 */
        for (uint32_t i = o->widthStartOffset; i < o->width + o->widthStartOffset; ++i) { /* :315 */
            render_pixel(&d, integrator, (int)i, (int)j, cam, pix, per_sample);
            pix += 3;
            if (per_sample) per_sample += 3 * (size_t)spp;
        }
    }
    if (stats) {
        __atomic_fetch_add(&stats->path_samples, local.path_samples, __ATOMIC_RELAXED);
        __atomic_fetch_add(&stats->density_lookups, local.density_lookups, __ATOMIC_RELAXED);
        __atomic_fetch_add(&stats->scatter_events, local.scatter_events, __ATOMIC_RELAXED);
        __atomic_fetch_add(&stats->shadow_rays, local.shadow_rays, __ATOMIC_RELAXED);
        __atomic_fetch_add(&stats->ray_queries, local.ray_queries, __ATOMIC_RELAXED);
        __atomic_fetch_add(&stats->step_limit_hits, local.step_limit_hits, __ATOMIC_RELAXED);
    }
    return VR_OK;
}

/* render + mythread + MultithreadingHelper::tryReservingChunk
 * (REF BaseIntegrator.cpp:131-190, 253-285; Utils/MultithreadingHelper.cpp:10-28) */
typedef struct render_job {
    const vro_scene* sc; const vr_options_pod* o; int integrator, rng_mode; uint64_t seed;
    float* fb; vr_render_stats* stats;
    uint32_t chunk, nchunks; uint32_t* next; pthread_mutex_t* mu; int rc;
} render_job;

static void* render_worker(void* arg)
{
    render_job* jb = (render_job*)arg;
    const vr_options_pod* o = jb->o;
    for (;;) {
        pthread_mutex_lock(jb->mu);
        uint32_t c = *jb->next;
        if (c < jb->nchunks) *jb->next = c + 1;
        pthread_mutex_unlock(jb->mu);
        if (c >= jb->nchunks) break;
        uint32_t off = c * jb->chunk;
        uint32_t from = o->heightStartOffset + off;                            /* :266 */
        uint32_t lim = off + jb->chunk;
        uint32_t to = o->heightStartOffset + (o->height < lim ? o->height : lim); /* :267-268 */
        int rc = vro_render_rows(jb->sc, o, jb->integrator, jb->rng_mode, jb->seed, from, to,
                                 jb->fb + 3 * (size_t)o->width * off, NULL, jb->stats); /* :276 */
        if (rc) jb->rc = rc;
    }
    return NULL;
}

int vro_render(const vro_scene* sc, const vr_options_pod* o, int integrator, int rng_mode, uint64_t seed,
               int nthreads, float* framebuffer, vr_render_stats* stats)
{
    int rc = scene_ok(sc, integrator);
    if (rc) return rc;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 512) nthreads = 512;
    if (stats) memset(stats, 0, sizeof *stats);
    uint32_t chunk = o->multiThreadedChunkSize > 0 ? o->multiThreadedChunkSize
                                                   : (o->height / (uint32_t)nthreads); /* :145-147 */
    if (chunk == 0) chunk = 1;
    uint32_t nchunks = (o->height + chunk - 1) / chunk;                        /* :154 */
    uint32_t next = 0;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
    render_job* jobs = (render_job*)malloc(sizeof(render_job) * (size_t)nthreads);
    for (int t = 0; t < nthreads; ++t) {
        render_job jb = { sc, o, integrator, rng_mode, seed, framebuffer, stats, chunk, nchunks, &next, &mu, 0 };
        jobs[t] = jb;
        pthread_create(&th[t], NULL, render_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); if (jobs[t].rc) rc = jobs[t].rc; }
    free(th); free(jobs);
    return rc;
}

/* ------------------------------------------------------------------------------------ */
/* post-process (REF BaseIntegrator.cpp:199-220 tonemap, :11-30 saveFile)                */
/* ------------------------------------------------------------------------------------ */

float vro_tonemap_reinhard_extended(float* rgb, uint64_t npixels)
{
    float maxWhite = 0.0f;                                                     /* :199-209 */
    for (uint64_t i = 0; i < 3 * npixels; ++i) if (rgb[i] > maxWhite) maxWhite = rgb[i];
    float maxWhite2 = maxWhite * maxWhite;                                     /* :211 */
    for (uint64_t i = 0; i < 3 * npixels; ++i) {                               /* :213-220 */
        float p = rgb[i];
        float numerator = p * ((p / maxWhite2) + 1.0f);
        rgb[i] = numerator / (p + 1.0f);
    }
    return maxWhite;
}

/* REF BaseIntegrator.cpp:19-25: clamp to [0,1], *255 truncated, row 0 of the framebuffer is
 * the TOP image row (FreeImage's y=height-1).  out is top-down RGB8. */
void vro_to_rgb8(const float* rgb, uint32_t width, uint32_t height, uint8_t* out)
{
    for (uint64_t i = 0; i < (uint64_t)width * height; ++i)
        for (int c = 0; c < 3; ++c)
            out[3 * i + c] = (uint8_t)(255 * u_clamp(0, 1, rgb[3 * i + c]));
}

/* REF Utils/XMLManager.cpp:39-107 (defaults of a config with no keys) */
void vro_options_defaults(vr_options_pod* o)
{
    memset(o, 0, sizeof *o);
    o->sigma_s = 0.5f; o->sigma_a = 0.5f;
    o->rayPerPixelCount = 1;
    o->maxDepth = 4;
    o->width = 400; o->widthReference = 400;
    o->height = 400; o->heightReference = 400;
    o->multiThreadedChunkSize = 20;
    o->fov = 90.0f;
    o->stepSizeMin = 0.01f; o->stepSizeMax = 1.0f; o->stepSizeMultiplier = 1.0f;
    o->lightRayDensityMultiplier = 1.0f; o->shadowRayDensityMultiplier = 1.0f;
}
