/*
 * vr_oracle.h — CPU oracle for the volume-renderer integrator hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's algorithm
 * (emiliano-luna/volume-renderer, /root/reference/VolumeRenderer/RayTracer/...).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as
 * the checker / the CPU baseline — never as a product path.  The product (libvr_backend.so)
 * neither links nor calls anything in this directory.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - function level: PINNED against the reference's own translation units compiled where
 *     they lie (oracle/_ref, built by oracle/Makefile: Utils.cpp, ONB.cpp, PhaseFunction.cpp,
 *     DirectionSampler.cpp, RandomGenerator.cpp, PerlinNoiseSampler.cpp and the header-only
 *     NanoVDB Ray / ReadAccessor / GridBuilder); vectors committed under tests/golden/.
 *   - castRay level (the four integrators' loops): PARITY UNPINNED.  The reference's
 *     integrator TUs include <Windows.h>/Process.h and link Embree + FreeImage, none of
 *     which exist in this image, and stand-ins are not allowed; the reference holds no test
 *     or golden vector for them.  The loops below follow the cited lines one to one.
 */
#ifndef VR_ORACLE_H
#define VR_ORACLE_H

#include "../include/vr_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* RNG modes for the render entry points */
enum {
    VRO_RNG_PHILOX    = 0, /* Philox4x32-10 keyed (pixel, sample, bounce): the GPU path's stream */
    VRO_RNG_REFERENCE = 1, /* mt19937 seeded per row chunk + uniform_real_distribution<float>
                              (Utils/RandomGenerator.cpp:4-14, BaseIntegrator.cpp:298) */
    VRO_RNG_PHILOX_GEOM = 2 /* VRO_RNG_PHILOX, and the delta tracker draws one exponential clock per walk instead of two
                               numbers per step wherever every free path is clamped to stepSizeMin (the FAST kernel's GEOM
                               variant; everything else draws as in VRO_RNG_PHILOX) */
};
/* Transcendental modes (logf/expf/sinf/cosf inside castRay) */
enum {
    VRO_MATH_LIBM   = 0, /* the C library, as the reference calls it */
    VRO_MATH_SHARED = 1  /* the polynomial kernels the HIP "exact" build also uses, so that
                            oracle and GPU agree bit for bit on the same Philox stream */
};

/* Oracle-only integrator: delta tracking over per-cell majorants (the estimator of VR_MODE_FAST_LOCAL / VR_MODE_EXACT_LOCAL;
 * no reference counterpart — vr_oracle.c, cast_ray_delta_local).  vro_local_majorants_build(grid, cell_log) first. */
enum { VRO_INTEGRATOR_DELTA_LOCAL = 100, VRO_INTEGRATOR_MESH_DELTA_LOCAL = 101 /* the same bounded by the mesh (meshDeltaTracking in the LOCAL modes) */ };
/* Study switches of the reference-faithful delta tracker (vr_oracle.c, cast_ray_delta); 0 = the reference as written */
enum { VRO_DELTA_LOOKUP_AFTER_STEP = 1, VRO_DELTA_CLAMP_ENTRY = 2 };

typedef struct vro_scene {
    const vr_grid_desc* density;     /* may be NULL */
    const vr_grid_desc* temperature; /* may be NULL */
    const vr_mesh_desc* mesh;        /* may be NULL */
} vro_scene;

void vro_set_math_mode(int mode);
int  vro_get_math_mode(void);
void vro_set_delta_variant(int bits);
int  vro_get_delta_variant(void);
int  vro_local_auto_cell_log(const vr_grid_desc* g);
int  vro_local_majorants_build(const vr_grid_desc* g, int cell_log);
void vro_local_majorants_free(void);

/* ---- function-level pieces (each cites the reference in vr_oracle.c) ---- */
void     vro_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
uint32_t vro_xoshiro128p_next(uint32_t s[4]);   /* the per-walk stream generator (state = one Philox block) */
void vro_xoshiro128p_next2(uint32_t s[4], uint32_t out[2]);   /* an iteration's two words: one transition, s0 + s3 and s1 + s2 */
void     vro_mt19937_seed(uint32_t* state625, uint32_t seed);
uint32_t vro_mt19937_next(uint32_t* state625);
float    vro_reference_get_float(uint32_t* state625, float lo, float hi);
void     vro_minstd_canonical_double(uint32_t* state, double* out);

float vro_shared_logf(float x);
float vro_shared_expf(float x);
float vro_shared_sinf(float x);
float vro_shared_cosf(float x);

void  vro_normalize(const float v[3], float out[3]);
float vro_clamp(float lo, float hi, float v);
void  vro_rotate(float pitch, float roll, float yaw, float p[3]);
void  vro_assign_point_to_quadrant(int i, int total, float out[3]);
/* camera constants of renderPartial: out = {scale, imageAspectRatio, pixelWidth, pixelHeight} */
void  vro_camera_constants(const vr_options_pod* o, float out[4]);
/* direction of sub-sample s of absolute pixel (i, j), before castRay */
void  vro_primary_direction(const vr_options_pod* o, int i, int j, uint32_t s, float dir[3]);

/* nanovdb::Ray<float>(eye, dir).worldToIndexF(grid) then clip(bbox).
 * out = {eye[3], dir[3], t0, t1}; returns 1 on hit. */
int   vro_world_ray_to_index_clip(const vr_grid_desc* g, const float eye[3], const float dir[3], float out[8]);
/* nanovdb::Ray<float>(eye, dir) (index space, default times) then clip(bbox) */
int   vro_index_ray_clip(const vr_grid_desc* g, const float eye[3], const float dir[3], float out[8]);
/* acc.getValue(Coord::Floor(p)) */
float vro_grid_value_at(const vr_grid_desc* g, const float p[3]);
void  vro_index_to_world(const vr_grid_desc* g, const float* p_xyz, int n, float* out_xyz);
float vro_grid_max(const vr_grid_desc* g);

float vro_henyey_greenstein(float g, float cos_theta);
void  vro_sample_henyey_greenstein(float g, const float dir[3], float xi1, float xi2, float out[3]);

float vro_perlin_noise(float x, float y, float z);
float vro_perlin_eval_density(const float p[3]);
/* BASELINE config 2-4 cloud: n^3 voxels, bbox centred on the origin, k fastest */
void  vro_make_cloud(int n, float radius_voxels, float* voxels, int nthreads);

/* closest hit over all triangles; returns prim id or -1, t in *t_out */
int   vro_intersect_closest(const vr_mesh_desc* m, const float org[3], const float dir[3], float* t_out);

/* ---- castRay level ---- */
/* One path sample of absolute pixel (i, j), sub-sample s, Philox stream. */
int vro_cast_ray_sample(const vro_scene* sc, const vr_options_pod* o, int integrator,
                        uint64_t seed, int i, int j, uint32_t s, float rgb[3],
                        vr_render_stats* stats);

/* renderPartial (BaseIntegrator.cpp:287-325): absolute rows [from_row, to_row), writes
 * width*(to_row-from_row) RGB triples to pix.  per_sample (optional) receives every
 * castRay result, laid out [row][col][s][3]. */
int vro_render_rows(const vro_scene* sc, const vr_options_pod* o, int integrator,
                    int rng_mode, uint64_t seed, uint32_t from_row, uint32_t to_row,
                    float* pix, float* per_sample, vr_render_stats* stats);

/* render (BaseIntegrator.cpp:131-190) minus tonemap/save: nthreads workers popping
 * multiThreadedChunkSize-row chunks. */
int vro_render(const vro_scene* sc, const vr_options_pod* o, int integrator,
               int rng_mode, uint64_t seed, int nthreads, float* framebuffer,
               vr_render_stats* stats);

/* post (BaseIntegrator.cpp:199-220, :11-30) */
float vro_tonemap_reinhard_extended(float* rgb, uint64_t npixels);
void  vro_to_rgb8(const float* rgb, uint32_t width, uint32_t height, uint8_t* out);

void vro_options_defaults(vr_options_pod* o);

#ifdef __cplusplus
}
#endif
#endif
