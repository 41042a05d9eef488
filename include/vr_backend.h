/*
 * vr_backend.h — C-ABI of the MI355X integrator backend (libvr_backend.so).
 *
 * Drop-in boundary for the per-pixel integrator hot path of emiliano-luna/volume-renderer.
 * Each entry point cites the reference interface it replaces; paths are relative to
 * /root/reference/VolumeRenderer/RayTracer/.  Plain pointers and sizes only; every function
 * returns VR_OK (0) or a negative VR_ERR_* (vr_types.h) and never throws or exits — the
 * reference's own convention is exit()/throw (SceneLoader.cpp:90,213), which cannot cross a
 * C boundary.  INTEGRATION.md shows the reference-side binding.
 *
 * A vr_backend is bound to one GPU and must be driven from one host thread at a time.
 * Inputs handed to vr_backend_set_* are copied to HBM before the call returns.
 */
#ifndef VR_BACKEND_H
#define VR_BACKEND_H

#include "vr_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vr_backend vr_backend;

/* Arithmetic mode of the integrator kernels. */
enum {
    /* Polynomial log/exp/sin/cos shared with the CPU oracle, one Philox stream per
     * (pixel, sample, bounce), one density lookup per step: results are bit-identical to
     * oracle/vr_oracle.c in VRO_MATH_SHARED + VRO_RNG_PHILOX mode.  The parity mode. */
    VR_MODE_EXACT = 0,
    /* Same algorithm and random streams, hardware transcendentals (v_log_f32, v_exp_f32,
     * v_sin_f32, v_cos_f32; in the trackers also v_rcp / v_sqrt / v_rsq where EXACT divides and
     * takes roots correctly rounded) and empty-space stepping batched over cubes of empty voxels:
     * differs from EXACT only by rounding.  The production mode. */
    VR_MODE_FAST = 1,
    /* EXACT arithmetic with FAST's batched empty-space stepping: must also be bit-identical to
     * the oracle; exists so that tests can pin the batching on its own. */
    VR_MODE_EXACT_SKIP = 2,
    /* FAST, and deltaTracking (and meshDeltaTracking) sampled against PER-CELL majorants (a coarse majorant grid walked by a 3-D DDA)
     * instead of the reference's single global one (IntegratorDeltaTracking.cpp:44-47).  Opt-in, and NOT the
     * reference's estimator: it is textbook delta tracking (the density is looked up at the tentative collision,
     * not one free path earlier), offered only where the reference's own parameters make it a plain delta
     * tracker — stepSizeMultiplier 1, stepSizeMin <= 0.01, stepSizeMax >= 1e4, no importance-sampling output —
     * else VR_ERR_UNSUPPORTED.  Statistically equal to FAST up to the smear of one free path; every other
     * integrator renders exactly as in VR_MODE_FAST. */
    VR_MODE_FAST_LOCAL = 3,
    /* The local-majorant delta tracker in EXACT arithmetic: bit-identical (image and counters) to the oracle's
     * restatement of that estimator (oracle/vr_oracle.c, cast_ray_delta_local, VRO_MATH_SHARED + VRO_RNG_PHILOX).
     * The parity mode of VR_MODE_FAST_LOCAL; every other integrator renders as in VR_MODE_EXACT_SKIP. */
    VR_MODE_EXACT_LOCAL = 4
};

/* ---- life cycle -------------------------------------------------------------------------
 * Replaces: SceneLoader::initializeDevice / rtcReleaseDevice (SceneLoader.cpp:25-34,
 * SceneRenderer.cpp:27-28) — the reference's only "device" object. */
int  vr_backend_create(int hip_device_ordinal, vr_backend** out);
void vr_backend_destroy(vr_backend* b);
/* Message of the last failing call on this backend ("" if none).  b may be NULL: then the
 * message of the last failing vr_backend_create of this thread. */
const char* vr_backend_last_error(const vr_backend* b);

/* ---- scene ------------------------------------------------------------------------------
 * Replaces the SceneInfo members the integrators read (Utils/Types.h:129-146), filled by
 * SceneLoader::initializeScene (SceneLoader.cpp:43-233). */

/* densityGrid / temperatureGrid + densityExtrema + gridBoundingBox (Types.h:141-145;
 * SceneLoader.cpp:207-230).  The voxels are re-laid out in HBM as 8x8x8 bricks and a
 * skip grid (every empty voxel's distance to the nearest non-empty one) is built.  temperature may be NULL. */
int vr_backend_set_density(vr_backend* b, const vr_grid_desc* density, const vr_grid_desc* temperature);

/* Fills the density grid on the device with the synthetic cloud of BASELINE configs 2-4:
 * n^3 voxels (n a multiple of 8), voxel size 1, bbox centred on the origin, Perlin fbm of
 * Utils/PerlinNoiseSampler.cpp:46-68 under a radial smoothstep.  No reference counterpart
 * (its fields come from .nvdb files that are absent from the checkout). */
int vr_backend_generate_cloud(vr_backend* b, int n, float radius_voxels);

/* Replaces the Map of the grid that is resident (world = voxel_size * index + origin): places the
 * generated cloud over a mesh for the mesh-bounded trackers.  The reference reads the Map from the
 * .nvdb file (grid->map(), used through worldToIndexF, nanovdb/util/Ray.h:257-267). */
int vr_backend_set_density_map(vr_backend* b, float voxel_size, const float origin_xyz[3]);

/* Reads the bricked density back as a dense k-fastest array of dims[0]*dims[1]*dims[2]
 * floats (for checks); dims/bbox_min may be NULL. */
int vr_backend_read_density(vr_backend* b, float* voxels, uint64_t capacity_floats, int32_t dims[3], int32_t bbox_min[3]);
int vr_backend_density_max(vr_backend* b, float* max_value);

/* scene + primitives + materials[].dissolve (Types.h:130-133; SceneLoader.cpp:69-175).
 * Builds the BVH that replaces rtcIntersect1 (Utils/EmbreeHelper.cpp:4-36). */
int vr_backend_set_mesh(vr_backend* b, const vr_mesh_desc* mesh);

/* ---- knobs without a reference counterpart ---------------------------------------------- */
int vr_backend_set_mode(vr_backend* b, int vr_mode);          /* default VR_MODE_FAST */
int vr_backend_set_seed(vr_backend* b, uint64_t philox_key);  /* default 0x5EED0001   */
/* Cap on the per-sample staging buffer in HBM (bytes); samples are rendered in passes that
 * fit (the buffer grows to what a frame needs, never beyond the cap).  Default 32 GiB. */
int vr_backend_set_sample_buffer_bytes(vr_backend* b, uint64_t bytes);

/* ---- the hot path -----------------------------------------------------------------------
 * vr_backend_render_rows replaces BaseIntegrator::renderPartial(Vec3f* orig, Vec3f* pix,
 * ThreadInfo*, const Options&, SceneInfo*) (integrators/BaseIntegrator.h:67, .cpp:287-325)
 * including the virtual castRay it calls (BaseIntegrator.h:65) for the integrators of
 * vr_integrator_id: absolute rows [from_row, to_row) (ThreadInfo::fromHeight/toHeight, i.e.
 * heightStartOffset already added), all columns [widthStartOffset, widthStartOffset+width);
 * writes width*(to_row-from_row) RGB float triples, row-major, first row first, into
 * caller-owned HOST memory (the reference's `Vec3f* pix`). */
int vr_backend_render_rows(vr_backend* b, const vr_options_pod* o, int integrator,
                           uint32_t from_row, uint32_t to_row, float* pix_rgb);

/* Replaces BaseIntegrator::render (BaseIntegrator.h:69, .cpp:131-190) up to the end of its
 * timed span (:192): the whole width x height framebuffer, before tonemap and save. */
int vr_backend_render(vr_backend* b, const vr_options_pod* o, int integrator, float* framebuffer_rgb);

/* The same work with the output left in HBM, for the multi-GPU tile scheduler (the
 * reference's chunk queue, BaseIntegrator.cpp:253-285 + Utils/MultithreadingHelper.cpp:10-28,
 * hands out row chunks; here rank r of N takes rows row_first + k*row_stride):
 * renders rows heightStartOffset + row_first + k*row_stride, k = 0..row_count-1, into
 * d_rgb = DEVICE pointer to row_count*width*3 floats.  Asynchronous on the backend's stream
 * (nothing is copied from host memory and the call does not wait for earlier work on the stream,
 * so consecutive frames or row sets queue up behind each other); vr_backend_synchronize waits for it.
 * The backend's stream is its own, non-blocking one: work that other streams (torch's, the
 * caller's) have queued on d_rgb must be complete before this call, and the caller must
 * vr_backend_synchronize before another stream reads d_rgb. */
int vr_backend_render_row_set_device(vr_backend* b, const vr_options_pod* o, int integrator,
                                     uint32_t row_first, uint32_t row_stride, uint32_t row_count,
                                     void* d_rgb);
int vr_backend_synchronize(vr_backend* b);

/* vr_backend_render_row_set_device for callers that cannot allocate.  NO COUNTERPART IN THE
 * REFERENCE: it exists for callers that cannot call hipMalloc — tight serving loops, harnesses
 * that forbid allocation, code that must not disturb a stream capture elsewhere (the call itself
 * has not been exercised UNDER capture: it reads back nothing, but its launches go to the
 * backend's own stream).  Same arguments, same frame (bit for bit)
 * and same asynchrony as vr_backend_render_row_set_device, but the call allocates and frees
 * nothing, creates no event and copies nothing from host memory: work counters, statistics
 * counters, the trackers' parameter block, the sub-sample offsets and the per-sample staging
 * buffer all live in one static device array of 129 MiB that every GPU the library runs on holds
 * until the process ends.  Limits (VR_ERR_INVALID_ARG with a message beyond them):
 * row_count*width <= 11 184 810 pixels per call (128 MiB of staging / 12 B; a 4K frame is 8.3 M
 * pixels; more sub-samples than fit are rendered in several passes) and rayPerPixelCount <= 65 536.
 * Every call on one GPU shares that GPU's array — also calls on different vr_backend handles — so
 * calls that overlap on one GPU race: wait (vr_backend_synchronize) for one before issuing the
 * next from another handle; calls on one handle are ordered by its stream.  No kernel time is
 * measured (vr_backend_last_stats reports the counters and kernel_ms = 0). */
int vr_backend_render_row_set_device_noalloc(vr_backend* b, const vr_options_pod* o, int integrator,
                                             uint32_t row_first, uint32_t row_stride, uint32_t row_count,
                                             void* d_rgb);

/* Counters of the last render call (vr_types.h) and the HIP-event time of its integrator
 * kernel launches in milliseconds (sum over passes), measured on the backend's stream. */
int vr_backend_last_stats(vr_backend* b, vr_render_stats* stats, float* kernel_ms, uint32_t* kernel_launches);

/* ---- post-process -----------------------------------------------------------------------
 * Replaces the max-white scan + extended Reinhard of BaseIntegrator::render
 * (BaseIntegrator.cpp:199-220) on a DEVICE buffer of npixels RGB triples, in place.
 * max_white_in < 0: compute the maximum from the buffer (single GPU); otherwise use the
 * given global maximum (multi-GPU: all-reduce MAX the per-rank maxima first).
 * max_white_out may be NULL.  Both run on the backend's own stream and wait for it before they
 * return: a d_rgb that another stream produced (an RCCL gather on torch's stream) must be complete
 * before the call. */
int vr_backend_tonemap_device(vr_backend* b, void* d_rgb, uint64_t npixels, float max_white_in, float* max_white_out);
int vr_backend_max_channel_device(vr_backend* b, const void* d_rgb, uint64_t npixels, float* max_out);
/* Replaces the pixel loop of BaseIntegrator::saveFile (BaseIntegrator.cpp:19-25): clamp to
 * [0,1], *255 truncated; out = top-down RGB8 (the reference's flip puts framebuffer row 0 at
 * the top of the PNG).  Host buffers. */
int vr_rgb_to_u8(const float* rgb, uint32_t width, uint32_t height, uint8_t* out_rgb8);

/* ---- one frame over several GPUs ---------------------------------------------------------
 * Replaces the fan-out INSIDE BaseIntegrator::render (BaseIntegrator.cpp:143-181: N worker threads
 * drawing row chunks from Utils/MultithreadingHelper.cpp:10-28, joined before the timed span ends at
 * :192): a vr_backend_group holds one vr_backend (own stream, own copy of the scene) per listed GPU
 * and is driven from ONE host thread in ONE process.  vr_group_render gives device i of N the rows
 * i, i+N, i+2N, ... of the frame, launches all devices before it waits for any, and gathers the row
 * blocks into the caller's HOST framebuffer (each device copies its rows straight to their interleaved
 * places, so the frame is bit-identical for every N: the random streams are keyed by absolute pixel
 * and sample).  vr_group_render_device gathers into DEVICE memory of the group's first GPU instead
 * (peer copies over xGMI; the frame can then be tonemapped there with vr_backend_tonemap_device on
 * vr_group_backend(g, 0)).  A device ordinal may be listed more than once (a rehearsal of N > 1 on a
 * box with one GPU; never a measurement).  The multi-process way — one rank per GPU, RCCL
 * all_gather — is volume_renderer_amd/scheduler.py over the same row sets. */
typedef struct vr_backend_group vr_backend_group;
int  vr_group_create(const int* hip_device_ordinals, uint32_t ndevices, vr_backend_group** out);
void vr_group_destroy(vr_backend_group* g);
uint32_t    vr_group_size(const vr_backend_group* g);
vr_backend* vr_group_backend(vr_backend_group* g, uint32_t index);   /* borrowed; NULL when out of range */
const char* vr_group_last_error(const vr_backend_group* g);          /* g may be NULL: last failing vr_group_create */
/* the scene and the knobs, replicated on every device of the group (288 GB of HBM each) */
int vr_group_set_density(vr_backend_group* g, const vr_grid_desc* density, const vr_grid_desc* temperature);
int vr_group_generate_cloud(vr_backend_group* g, int n, float radius_voxels);
int vr_group_set_density_map(vr_backend_group* g, float voxel_size, const float origin_xyz[3]);
int vr_group_set_mesh(vr_backend_group* g, const vr_mesh_desc* mesh);
int vr_group_set_mode(vr_backend_group* g, int vr_mode);
int vr_group_set_seed(vr_backend_group* g, uint64_t philox_key);
/* BaseIntegrator::render up to :192 over all devices of the group; framebuffer_rgb = width*height*3
 * floats of HOST memory (pageable is fine: it is page-locked for the duration of the call, or the rows travel
 * through a pinned staging buffer of the group, so that the devices' copies run as DMA side by side).  Returns after every device's rows have arrived — also on failure: nothing is in flight
 * when the call returns. */
int vr_group_render(vr_backend_group* g, const vr_options_pod* o, int integrator, float* framebuffer_rgb);
/* The same with the frame gathered in HBM: d_rgb = width*height*3 floats on the group's first GPU. */
int vr_group_render_device(vr_backend_group* g, const vr_options_pod* o, int integrator, void* d_rgb);
/* How the last gathers went, for diagnosing a slow one: out[0] / out[1] = devices (other than the first) with / without
 * peer access to the group's first GPU (without it vr_group_render_device's copies go through the host), out[2] = the way
 * the last vr_group_render reached the caller's framebuffer: 1 = page-locked for the call (asynchronous copies straight
 * to the rows' places), 2 = through the group's own pinned staging buffer (asynchronous copies, then a host memcpy per
 * device as it finishes; taken where the platform refuses hipHostRegister), 0 = strided copies into pageable memory. */
int vr_group_gather_info(const vr_backend_group* g, uint32_t out[3]);
/* Counters summed over the devices; kernel_ms = the slowest device's integrator kernel time (the
 * devices run side by side), kernel_launches = the sum. */
int vr_group_last_stats(vr_backend_group* g, vr_render_stats* stats, float* kernel_ms, uint32_t* kernel_launches);

/* sizeof() of the boundary structs as compiled, for binding self-checks */
void vr_abi_sizes(uint32_t out[4]); /* options, grid_desc, mesh_desc, render_stats */

#ifdef __cplusplus
}
#endif
#endif /* VR_BACKEND_H */
