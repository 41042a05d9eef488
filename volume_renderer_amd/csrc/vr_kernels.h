// vr_kernels.h — the integrator kernels (gfx950, wave64).
//
//   vr_track_kernel<INTEG, EXACT, SKIP, IS, TEMP, SPAN>
//                                  ratio tracking / delta tracking through the bricked grid
//                                  (IntegratorRatioTracking.cpp:11-283, IntegratorDeltaTracking.cpp:11-176)
//   vr_march_kernel<KIND, EXACT>   walks through mesh-bounded media: the two homogeneous fixed-step marchers
//                                  (IntegratorHomogeneousRayMarcherNEE.cpp:10-118, ...Improved.cpp:11-149), the
//                                  Perlin-noise marcher (IntegratorHeterogeneousPerlinNoise.cpp:12-168) and the
//                                  analytic densitySampling (IntegratorDensitySampling.cpp:7-78)
//   vr_gridmarch_kernel<EMISSION, EXACT> fixed-step marchers through the grid
//                                  (IntegratorNanoVDBSimple.cpp:10-95, IntegratorNanoVDBEmission.cpp:10-110)
//   vr_resolve_kernel              per-pixel ordered sum over samples (BaseIntegrator.cpp:71,92,108)
//
// Work decomposition: one path sample (one castRay call) per lane.  Sample (s, r, x) owns slot
// ((s * nrows + r) * width + x) of the per-sample buffer [s][r][x][rgb]; vr_resolve_kernel sums the
// slots of a pixel in sample order, so the image does not depend on which lane ran which path.
// Path lengths differ by orders of magnitude (a background pixel ends at the bbox clip, a core pixel
// takes 10^4 steps), so the tracking kernel is persistent: a lane whose path has ended takes the next
// item of its wave's chunk (one atomic on its strip's counter per chunk of 128 items) instead of idling
// until the wave's longest path ends; items go out as 8x8-pixel tiles, sub-samples of a tile together,
// tile rows from the middle of the frame outwards, one strip of tile columns per XCD (see the refill).
// The mesh marchers take 8x8-pixel tiles in a grid-stride loop, the grid marchers draw them from per-strip
// counters.
#pragma once
#include "vr_device.h"

namespace vr {

enum { INTEG_RATIO = 2, INTEG_DELTA = 3 };

struct Counters {
    unsigned long long path_samples, density_lookups, scatter_events, shadow_rays, ray_queries, step_limit_hits;
    unsigned long long wave_trips, wave_trip_lanes, event_phases;
    unsigned long long voxel_loads;   // lookups that issued a load (the rest were served from the lane's register)
};

// Camera block: renderPartial's per-chunk constants + renderRay's per-pixel arithmetic
// (BaseIntegrator.cpp:33-57, 287-312).  rot = the matrix Utils::rotate builds (Utils.cpp:34-53).
struct CameraDev {
    float    pos[3];
    float    rot[9];
    int32_t  has_rot;
    float    scale, aspect, pixel_w, pixel_h;
    uint32_t ref_w, ref_h;   // widthReference/heightReference (or width/height when 0)
    uint32_t x0, y0;         // widthStartOffset, heightStartOffset
    uint32_t width;          // columns rendered
    uint32_t row_first, row_stride, nrows;
    uint32_t spp, s_first, s_count;
    const float2* offsets;   // assignPointToQuadrant(s, spp) for s < spp
};

struct WorkDev {
    uint32_t           nitems;      // s_count * nrows * width
    uint32_t           chunk;       // items a wave takes from the counter at a time (tracking kernel)
    uint32_t*          counter;     // next item
    float*             samples;     // [s_count][nrows][width][3]
    Counters*          counters;
    uint32_t           key0, key1;  // Philox key
    uint32_t           step_limit;
};

// primary ray of (abs column i, abs row j, sub-sample s)                  BaseIntegrator.cpp:35-64
// EXACT: the reference's arithmetic (x and y in double, correctly rounded normalisation); otherwise the same
// formula in float with a hardware reciprocal square root — one ulp on a direction, for the FAST trackers only
template <bool EXACT = true>
__device__ __forceinline__ f3 primary_direction(const CameraDev& c, uint32_t i, uint32_t j, uint32_t s)
{
    float x, y;
    if (EXACT) {
        x = (float)((2 * ((int)i + 0.5) / (float)c.ref_w - 1) * c.aspect * c.scale);
        y = (float)((1 - 2 * ((int)j + 0.5) / (float)c.ref_h) * c.scale);
    } else {
        x = ((2.0f * ((float)i + 0.5f)) * __builtin_amdgcn_rcpf((float)c.ref_w) - 1.0f) * c.aspect * c.scale;
        y = (1.0f - (2.0f * ((float)j + 0.5f)) * __builtin_amdgcn_rcpf((float)c.ref_h)) * c.scale;
    }
    f3 dn = Math<EXACT>::normalize_(mk3(x, y, -1));
    float2 off = c.offsets[s];
    f3 d = mk3(dn.x + off.x * c.pixel_w, dn.y + off.y * c.pixel_h, dn.z + 0.0f);
    if (c.has_rot) {
        float px = d.x, py = d.y, pz = d.z;
        d.x = c.rot[0] * px + c.rot[1] * py + c.rot[2] * pz;
        d.y = c.rot[3] * px + c.rot[4] * py + c.rot[5] * pz;
        d.z = c.rot[6] * px + c.rot[7] * py + c.rot[8] * pz;
    }
    return d;
}

__device__ __forceinline__ void store_sample(float* samples, uint32_t item, f3 c)
{
    float* o = samples + 3ull * item;
    o[0] = c.x; o[1] = c.y; o[2] = c.z;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, o), hi = __shfl_xor((uint32_t)(v >> 32), o);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

// =============================================================================================
// closest hit over the mesh (stand-in for rtcIntersect1)
// =============================================================================================
// Threaded BVH in depth-first order: a node that is hit continues at node+1 (its first child, or
// for a leaf the next subtree), a node that is missed continues at `skip`.  No stack.  MeshDev::nodes
// holds eight such layouts of the same tree, one per sign octant of the ray direction, each visiting
// the child nearer to its rays first.
// A node is 32 bytes — ONE aligned s_load_dwordx8 of the wave-uniform walk, never straddling a cache line.
struct BvhNode {
    float    lo[3], hi[3];
    uint32_t skip;
    uint32_t leaf;           // leaf: count << 24 | first — triangles [first, first + count); inner: 0
};
#define VR_BVH_FIRST(leaf) ((leaf) & 0xffffffu)
#define VR_BVH_COUNT(leaf) ((leaf) >> 24)
struct TriDev {
    float    v0[3], e1[3], e2[3];
    uint32_t prim;    // original primitive id (tie-break + prim_to_shape index)
};
struct MeshDev {
    const BvhNode*  nodes;
    const TriDev*   tris;
    const uint32_t* prim_to_shape;
    const float*    shape_dissolve;
    const float*    shape_diffuse;    // nshapes * 3 or nullptr
    uint32_t        nnodes;
};
// Moeller-Trumbore with the operation order of the oracle's tri_hit (stand-in for rtcIntersect1,
// Utils/EmbreeHelper.cpp:4-36; tnear = 0, tfar = inf)
__device__ __forceinline__ bool tri_hit(const TriDev& tr, f3 org, f3 dir, float& t_out)
{
    float px = dir.y * tr.e2[2] - dir.z * tr.e2[1];
    float py = dir.z * tr.e2[0] - dir.x * tr.e2[2];
    float pz = dir.x * tr.e2[1] - dir.y * tr.e2[0];
    float det = tr.e1[0] * px + tr.e1[1] * py + tr.e1[2] * pz;
    if (det == 0.0f) return false;
    float inv = 1.0f / det;
    float tx = org.x - tr.v0[0], ty = org.y - tr.v0[1], tz = org.z - tr.v0[2];
    float u = (tx * px + ty * py + tz * pz) * inv;
    if (u < 0.0f || u > 1.0f) return false;
    float qx = ty * tr.e1[2] - tz * tr.e1[1];
    float qy = tz * tr.e1[0] - tx * tr.e1[2];
    float qz = tx * tr.e1[1] - ty * tr.e1[0];
    float v = (dir.x * qx + dir.y * qy + dir.z * qz) * inv;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = (tr.e2[0] * qx + tr.e2[1] * qy + tr.e2[2] * qz) * inv;
    if (!(t >= 0.0f)) return false;
    t_out = t;
    return true;
}

// closest hit; returns primitive id or -1
__device__ __forceinline__ int closest_hit(const MeshDev& m, f3 org, f3 dir, float& t_out)
{
    const float ix = 1.0f / dir.x, iy = 1.0f / dir.y, iz = 1.0f / dir.z;
    float    best_t = __builtin_huge_valf();
    uint32_t best_prim = 0xffffffffu;
    uint32_t n = 0;
    const BvhNode* nodes = m.nodes + ((dir.x < 0.0f ? 1u : 0u) | (dir.y < 0.0f ? 2u : 0u) | (dir.z < 0.0f ? 4u : 0u)) * m.nnodes;
    while (n < m.nnodes) {
        const BvhNode nd = nodes[n];
        // slab test; fminf/fmaxf drop NaNs (0 * inf), which keeps the test conservative
        float ax = (nd.lo[0] - org.x) * ix, bx = (nd.hi[0] - org.x) * ix;
        float ay = (nd.lo[1] - org.y) * iy, by = (nd.hi[1] - org.y) * iy;
        float az = (nd.lo[2] - org.z) * iz, bz = (nd.hi[2] - org.z) * iz;
        float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
        float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), best_t));
        if (tn <= tf) {
            for (uint32_t k = 0; k < VR_BVH_COUNT(nd.leaf); ++k) {
                const TriDev tr = m.tris[VR_BVH_FIRST(nd.leaf) + k];
                float        t;
                if (tri_hit(tr, org, dir, t) && (t < best_t || (t == best_t && tr.prim < best_prim))) {
                    best_t = t; best_prim = tr.prim;
                }
            }
            n = n + 1;
        } else {
            n = nd.skip;
        }
    }
    t_out = best_t;
    return best_prim == 0xffffffffu ? -1 : (int)best_prim;
}

// The same query for a wave whose rays are COHERENT (the marchers: an 8x8-pixel tile's primary rays, its light rays): the
// wave walks ONE node sequence — the union of its lanes' traversals, which is what a wave executes anyway — with the node
// index in a scalar register, so a node or a triangle is fetched ONCE per wave by the scalar unit (s_load through the
// constant address space: the BVH is read-only while a kernel runs) instead of 64 times by 64 lanes: the per-lane walk spent
// ~700 cycles per node visit on the L1 request path for ~100 cycles of arithmetic (profiles/r02_config5_marcher).
// Per lane nothing changes: a lane whose own slab test fails at node m sits out until the walk reaches m's skip target
// (`resume`; the order is depth-first, so everything in between is the subtree that lane would have skipped), and a leaf's
// triangles are tested only by lanes whose own slab test passed — the same triangles against the same best_t as
// closest_hit, hence the same result bit for bit.  Lanes of different direction octants use different node orders: the
// octants present in the wave are served one after the other (one for all light rays, rarely more than one for a tile).
// Must be called by all lanes that are active together; wave-level operations see exactly those.
typedef const uint32_t __attribute__((address_space(4)))* const_u32p;
__device__ __forceinline__ const_u32p as_constant(const void* p) { return (const_u32p)(unsigned long long)p; }
__device__ __forceinline__ int closest_hit_wave(const MeshDev& m, f3 org, f3 dir, float& t_out)
{
    const float ix = 1.0f / dir.x, iy = 1.0f / dir.y, iz = 1.0f / dir.z;
    float    best_t = __builtin_huge_valf();
    uint32_t best_prim = 0xffffffffu;
    const uint32_t oct = (dir.x < 0.0f ? 1u : 0u) | (dir.y < 0.0f ? 2u : 0u) | (dir.z < 0.0f ? 4u : 0u);
    const uint32_t nn = m.nnodes;
    bool pending = true;
    while (pending) {
        const uint32_t uoct = (uint32_t)__builtin_amdgcn_readfirstlane((int)oct);
        if (oct == uoct) {
            pending = false;
            const const_u32p nodes = as_constant(m.nodes) + (size_t)uoct * nn * 8u;
            const const_u32p tris = as_constant(m.tris);
            uint32_t n = 0;          // wave-uniform
            uint32_t resume = 0;     // this lane takes part from node `resume` on
            // (Fetching node n + 1 — where the walk goes whenever some lane hits — beside node n was measured: 30 % SLOWER.
            // Scalar loads return out of order, so the first use of either waits for both.)
            while (n < nn) {
                const const_u32p q = nodes + (size_t)n * 8u;
                const float lo0 = __uint_as_float(q[0]), lo1 = __uint_as_float(q[1]), lo2 = __uint_as_float(q[2]);
                const float hi0 = __uint_as_float(q[3]), hi1 = __uint_as_float(q[4]), hi2 = __uint_as_float(q[5]);
                const uint32_t skip = q[6], first = VR_BVH_FIRST(q[7]), count = VR_BVH_COUNT(q[7]);
                const bool act = n >= resume;
                float ax = (lo0 - org.x) * ix, bx = (hi0 - org.x) * ix;
                float ay = (lo1 - org.y) * iy, by = (hi1 - org.y) * iy;
                float az = (lo2 - org.z) * iz, bz = (hi2 - org.z) * iz;
                float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
                float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), best_t));
                const bool hit = act && tn <= tf;
                if (act && !hit) resume = skip;
                if (__ballot(hit) != 0ull) {
                    for (uint32_t k = 0; k < count; ++k) {
                        const const_u32p tq = tris + (size_t)(first + k) * (sizeof(TriDev) / 4u);
                        TriDev tr;
                        tr.v0[0] = __uint_as_float(tq[0]); tr.v0[1] = __uint_as_float(tq[1]); tr.v0[2] = __uint_as_float(tq[2]);
                        tr.e1[0] = __uint_as_float(tq[3]); tr.e1[1] = __uint_as_float(tq[4]); tr.e1[2] = __uint_as_float(tq[5]);
                        tr.e2[0] = __uint_as_float(tq[6]); tr.e2[1] = __uint_as_float(tq[7]); tr.e2[2] = __uint_as_float(tq[8]);
                        tr.prim = tq[9];
                        float t;
                        if (hit && tri_hit(tr, org, dir, t) && (t < best_t || (t == best_t && tr.prim < best_prim))) {
                            best_t = t; best_prim = tr.prim;
                        }
                    }
                    n = n + 1u;
                } else {
                    n = skip;
                }
                n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
            }
        }
    }
    t_out = best_t;
    return best_prim == 0xffffffffu ? -1 : (int)best_prim;
}
#ifndef VR_MARCH_WAVE_BVH
#define VR_MARCH_WAVE_BVH 1
#endif
__device__ __forceinline__ int closest_hit_march(const MeshDev& m, f3 org, f3 dir, float& t_out)
{
    return VR_MARCH_WAVE_BVH ? closest_hit_wave(m, org, dir, t_out) : closest_hit(m, org, dir, t_out);
}

// Mesh-bounded media for the trackers (VR_INTEGRATOR_MESH_*_TRACKING) — NO REFERENCE COUNTERPART: the reference's
// trackers clip to the grid bbox only (IntegratorDeltaTracking.cpp:37-39); SURVEY 8(d) config 5 defines the capability.
// The medium interval of a ray is found by the walk the reference's MARCHERS do before they march
// (IntegratorHomogeneousRayMarcherNEE.cpp:16-60, 104-108): closest hit; a hit on another object than the last one is
// an entry and moves the origin to hit + dir * 0.001; the next hit on the same object is the exit.  A ray that starts
// inside a medium (scattered rays, shadow rays) starts with that medium's object as "the last one".  Opaque shapes
// (dissolve <= 0) are black absorbers.  [s0, s1] is in units of |dir| from org.  (oracle: mesh_segment)
enum { SEG_NONE = 0, SEG_FOUND = 1, SEG_BLACK = 2 };
#ifndef VR_MESH_TRACK_WAVE_BVH
#define VR_MESH_TRACK_WAVE_BVH 0
#endif
__device__ __forceinline__ int mesh_segment(const MeshDev& m, f3 org, f3 dir, int inside_id, float& s0, float& s1, int& medium, uint32_t& n_queries)
{
    int   object = inside_id;
    float s_acc = 0.0f, start = 0.0f;
    for (int guard = 0; guard < 64; ++guard) {
        float t;
        n_queries++;
        const int prim = VR_MESH_TRACK_WAVE_BVH ? closest_hit_wave(m, org, dir, t) : closest_hit(m, org, dir, t);
        if (prim < 0) return SEG_NONE;
        const int previous = object;
        object = (int)m.prim_to_shape[prim];
        if (!(m.shape_dissolve[object] > 0)) return SEG_BLACK;
        if (previous == object) { s0 = start; s1 = s_acc + t; medium = object; return SEG_FOUND; }
        const f3 hit = add3(scale3(dir, t), org);
        org = add3(hit, scale3(dir, 0.001f));
        s_acc = (s_acc + t) + 0.001f;
        start = s_acc;
    }
    return SEG_NONE;
}
// cut [t0, t1] by the mesh interval of the matching world-space ray; scale = index parameter per world parameter
__device__ __forceinline__ int mesh_clip(const MeshDev& m, f3 worg, f3 wdir, float scale, int inside_id, int& medium, float& t0, float& t1,
                                         bool& empty, uint32_t& n_queries)
{
    float s0 = 0.0f, s1 = 0.0f;
    empty = false;
    const int rc = mesh_segment(m, worg, wdir, inside_id, s0, s1, medium, n_queries);
    if (rc != SEG_FOUND) return rc;
    const float a = s0 * scale, b = s1 * scale;
    if (a > t0) t0 = a;
    if (b < t1) t1 = b;
    if (t0 > t1) empty = true;
    return rc;
}
// Map::applyMapF (NanoVDB.h:2249)
__device__ __forceinline__ f3 index_to_world(const float* mat, const float* vec, f3 p)
{
    return mk3(fmaf(p.x, mat[0], fmaf(p.y, mat[1], fmaf(p.z, mat[2], vec[0]))), fmaf(p.x, mat[3], fmaf(p.y, mat[4], fmaf(p.z, mat[5], vec[1]))),
               fmaf(p.x, mat[6], fmaf(p.y, mat[7], fmaf(p.z, mat[8], vec[2]))));
}

// Parameters only the event phases read (path set-up, scatter, shadow start/end).  They live in
// device memory and are read with ordinary loads there, so that they do not occupy scalar
// registers across the step loop.
struct TrackCold {
    CameraDev cam;
    float     inv_mat[9], vec[3];
    float     bmin[3], bmax1[3];     // float(bbox.min), float(bbox.max + 1)
    float     g;
    float     light[3], light_color[3];
    float     emission[3], medium[3], background[3];
    uint32_t  max_depth;
    uint32_t  tile_h_log;           // work tiles are (64 >> tile_h_log) columns x (1 << tile_h_log) rows of the row set
    GridDev   temperature;
    // mesh-bounded trackers only (MESH kernels)
    MeshDev   mesh;
    float     mat[9];               // index -> world (Map::applyMapF), for the mesh queries of index-space rays
};
// Kernel argument: what the step loop reads.
struct TrackParams {
    const TrackCold* cold;
    WorkDev   work;
    GridDev   density;               // SKIP kernels: the skip grid (empty voxels hold -(empty-cube radius + 1))
    int32_t   round_max, round_events; // step-loop trips per round / waiting lanes that end a round early
    float     sigma_a, sigma_s, sigma_maj;
    float     t_min, t_max, step_mult;
    float     light_mult, shadow_mult;
};

// =============================================================================================
// ratio / delta tracking
// =============================================================================================
// Lane states.  The kernel alternates between a STEP LOOP, in which every walking lane runs the
// same code (a few iterations of the reference's while loop per trip, camera walk and shadow
// walk alike), and an EVENT PHASE, in which lanes that left the loop are served together: scatter
// events start their shadow walk, finished shadow walks add the light and pick the new direction,
// finished paths store their sample and take the next work item.  A lane that meets an event waits
// (a few iterations) for the phase instead of dragging the whole wave through the long, rare code.
enum {
    ST_NEED = 0, ST_WALK = 1, ST_EXHAUSTED = 2,
    // events, served by the event phase (every state >= ST_SCATTER)
    ST_SCATTER = 3, ST_SHADOW_DONE = 4, ST_FINISHED = 5, ST_ESCAPED = 6, ST_ABSORBED = 7,
    ST_COLLIDED = 8   // FLAT iterations: a real collision whose kind (absorption / scattering) the event phase works out
};
// FLAT iterations (FAST kernels): the step loop's iteration as straight-line code.  The scalar unit — ONE per CU, shared by
// its four SIMDs — is the kernel's other bottleneck: every divergent `if` costs s_and_saveexec / s_cbranch_execz / s_or exec
// plus the mask arithmetic around it, a trip was 310 vector and 230 scalar instructions, and that mix issues at 1.10 vector
// and 0.82 scalar instructions per CU cycle against 1.48 for vector code alone (tools/ubench_salu.hip).  So the iteration
// decides with selects, a walk that leaves its segment or collides only changes state, and the bookkeeping of a real
// collision (its kind, the scatter point, the budget) moves to the event phase, where all waiting lanes are served together.
#ifndef VR_FLAT_ITER
#define VR_FLAT_ITER 1
#endif
#ifndef VR_ONE_STREAM_EXP
#define VR_ONE_STREAM_EXP 0   // timing experiment: FAST walks go on drawing from the path's first stream (no Philox per walk)
#endif
// Wave priorities (s_setprio): a wave in the step loop is preferred by its SIMD's issue arbiter over a wave in the event phase, which
// serves a dozen lanes with long dependent chains (Philox, reciprocals, the lookup of a new ray) and fills the slots the walkers leave.
// Measured with 3 / 0 against no priorities: config 2 643.6 -> 634.2 ms (+1.5 %), config 3 as shipped +0.9 %, config 2 textbook +0.8 %,
// the reverse (event phase first) 0 to -1 %.  VR_EVENT_PRIO -1 = none.
#ifndef VR_EVENT_PRIO
#define VR_EVENT_PRIO 0
#endif
#ifndef VR_STEP_PRIO
#define VR_STEP_PRIO 3
#endif
// ... and within a trip the iterations (44 of 64 lanes busy) go before the empty-space and lookup blocks (10 and 22 lanes): another
// +0.7 % on config 2.  Only the kernels with sub-voxel steps (SPAN) set priorities: with voxel-sized steps (one iteration per trip,
// a lookup in every one) the same settings cost config 3 textbook 0.8 % + 0.9 %.
#ifndef VR_SERVICE_PRIO
#define VR_SERVICE_PRIO 1
#endif
#define VR_ROUND_MAX 48      // step-loop trips before the event phase runs regardless
#define VR_ROUND_EVENTS 12   // ... or as soon as this many lanes wait for it
// Iterations of the reference's loop per trip.  A trip ends with ONE empty-space service and ONE voxel
// lookup for the whole wave, so more iterations per trip amortise those — as long as lanes rarely leave
// their voxel (SPAN: sub-voxel steps), because a lane that does sits out the rest of the trip.
#ifndef VR_TRIP_ITERS_SPAN_RATIO
#define VR_TRIP_ITERS_SPAN_RATIO 3
#endif
#ifndef VR_TRIP_ITERS_SPAN_DELTA
#define VR_TRIP_ITERS_SPAN_DELTA 6
#endif
#ifndef VR_TRIP_ITERS_PLAIN
#define VR_TRIP_ITERS_PLAIN 1
#endif
// per-lane count of the voxel loads issued (vr_render_stats::voxel_loads); -DVR_NO_LOAD_COUNT builds measure its cost
#ifdef VR_NO_LOAD_COUNT
#define VR_COUNT_LOAD(n) ((void)0)
#else
#define VR_COUNT_LOAD(n) ((n)++)
#endif
// work hand-out of the tracking kernel: vertical strips of tile columns, one per XCD
#define VR_TRACK_STRIPS 8

// hide a pointer's provenance: loads through it stay ordinary vector loads inside the phase
template <typename T>
__device__ __forceinline__ const T* opaque(const T* p)
{
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+v"(v));
    return (const T*)v;
}

// Lane state that only the event phases (and the rare scatter / absorption branches of the step loop) touch
// lives in LDS, one dword per lane and variable: 15 of the ~110 registers the state would need, which is what
// lets the step loop keep the rest in 96.  These proxies make such a variable read and assign like a local.
struct LdsF3 {
    float* p;   // this lane's x; y and z follow at strides of 256 lanes
    __device__ __forceinline__ operator f3() const { return mk3(p[0], p[256], p[512]); }
    __device__ __forceinline__ LdsF3& operator=(f3 v) { p[0] = v.x; p[256] = v.y; p[512] = v.z; return *this; }
    __device__ __forceinline__ LdsF3& operator=(const LdsF3& o) { return *this = (f3)o; }
};

// The ray's geometry (origin, direction, and the two vectors of the voxel-span estimate) is read once per trip,
// by the lookup: with it in LDS as well a FAST kernel needs 73-80 registers and six waves fit a SIMD (26 KB of
// LDS per block, six blocks per CU).  Measured against five waves with the geometry in registers: +7 % where
// trips are long (sub-voxel steps) and for the ratio tracker; for the delta tracker with voxel-sized steps, which
// looks a voxel up in every iteration, -4 % while that kernel was bound by L2 misses and +6-8 % once the tiled
// work order had removed them.
#ifndef VR_GEO_IN_LDS
#define VR_GEO_IN_LDS(INTEG, EXACT, SPAN) (!(EXACT))
#endif
// Waves per SIMD the register allocator must allow (EXACT kernels: 98-123 VGPRs, four waves).
#ifndef VR_TRACK_WAVES
#define VR_TRACK_WAVES(INTEG, EXACT, SPAN) ((EXACT) ? 4 : (VR_GEO_IN_LDS(INTEG, EXACT, SPAN) ? 6 : 5))
#endif
template <bool IN_LDS>
struct GeoVec {
    typedef LdsF3 type;
    static __device__ __forceinline__ LdsF3 make(float* lane_slot) { return LdsF3{lane_slot}; }
};
template <>
struct GeoVec<false> {
    typedef f3 type;
    static __device__ __forceinline__ f3 make(float*) { return mk3(0, 0, 1); }
};

// SKIP: take runs of empty-space iterations at once (empty_run, vr_device.h) — same results bit
// for bit, fewer trips through the loop.
// IS: Options::useImportanceSampling — track the path pdf and divide the sample by max(0.1, pdf)
// (a compile-time switch: as a run-time flag it cost the common case 1.3 %).
// TEMP: a temperature grid is present (emission lookups); SPAN: steps are much shorter than a voxel,
// remember the voxel's t-interval.  Compile-time too: as run-time flags in the step loop they cost
// 2.5 % and 6.5 %.
// MESH: the walks are cut to the mesh's entry/exit interval (mesh_segment; event phases only).  Those kernels keep one
// more dword of lane state in LDS (the medium's object) and get a wave less per SIMD: closest_hit needs registers
// while the step loop's state is alive.
#ifndef VR_MESH_WAVES_LESS
#define VR_MESH_WAVES_LESS 1
#endif
// GEOM (FAST delta tracking, SKIP + SPAN, no IS; chosen by the host when EVERY free path is clamped to stepSizeMin, i.e.
// -ln(1e-5) * stepSizeMultiplier / sigma_maj <= stepSizeMin — the reference's shipped delta config,
// bunny-cloud-deltaTracking-configRenderer.xml:13): the walk is then a fixed-step march of stepSizeMin whose steps are
// independent Bernoulli trials, real collision with probability 1 - pNull(sigma) (IntegratorDeltaTracking.cpp:62-99), and
// sigma only changes with the voxel.  Instead of ~100 iterations per voxel the kernel takes a voxel's steps in ONE run:
// the walk carries an exponential clock E = -log2(u), drawn once per walk; a step in a voxel costs -log2(pNull), and the
// real collision is the step at which the clock runs out (P(no collision in steps 1..m) = prod pNull_i = P(E > sum cost_i):
// the same distribution as the reference's per-step draws, another stream).  A run is empty_run's arithmetic — t advances
// exactly as that many float additions would — with the step index of the collision solved for in closed form.  Empty
// voxels (cost 0, step 10 * stepSizeMin, the skip grid's cube) go through the same code.  The oracle restates the clock
// step by step (VRO_RNG_PHILOX_GEOM); EXACT modes keep the reference's draw per step.
template <int INTEG, bool EXACT, bool SKIP, bool IS, bool TEMP, bool SPAN, bool MESH = false, bool GEOM = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VR_TRACK_WAVES(INTEG, EXACT, SPAN) - (MESH && !(EXACT) ? VR_MESH_WAVES_LESS : 0), VR_TRACK_WAVES(INTEG, EXACT, SPAN) - (MESH && !(EXACT) ? VR_MESH_WAVES_LESS : 0))))
vr_track_kernel(const TrackParams p)
{
    static_assert(!GEOM || (INTEG == INTEG_DELTA && !EXACT && SKIP && !IS && SPAN), "GEOM: FAST delta tracking over the skip grid with voxel spans");
    typedef Math<EXACT> M;
    constexpr int  VR_TRIP_ITERS = SPAN ? (INTEG == INTEG_RATIO ? VR_TRIP_ITERS_SPAN_RATIO : VR_TRIP_ITERS_SPAN_DELTA) : VR_TRIP_ITERS_PLAIN;
    constexpr bool FLAT = VR_FLAT_ITER && !EXACT && !GEOM && !(INTEG == INTEG_RATIO && TEMP);
    const int      lane = threadIdx.x & 63;
    const uint32_t k0 = p.work.key0, k1 = p.work.key1;
    // FAST arithmetic: constants of the step loop folded once (wave-uniform)
    const float fast_c_ln = -0.6931471805599453f * p.step_mult;                 // len = log2(u) * fast_c_ln / denominator
    const float fast_c_cam = fast_c_ln / p.sigma_maj;
    const float fast_sig_t = p.sigma_a + p.sigma_s;                             // mu_t = sigma * fast_sig_t
    const float fast_sig_t2 = (p.sigma_a + p.sigma_s) * 1.4426950408889634f;    // optical depth in bits = len * sigma * fast_sig_t2
    const float fast_pa24 = p.sigma_a / p.sigma_maj * 16777216.0f, fast_ps24 = p.sigma_s / p.sigma_maj * 16777216.0f;

    // ---- per-lane path state (HandleIntersectionData, BaseIntegrator.h:21-50) ----
    uint32_t st = ST_NEED;
    bool     shadow = false;            // walking directLightningRayMarch
    __shared__ float s_cold[MESH ? 27 : 26][256];   // 26 KB per block: six blocks per CU fit the 160 KB of LDS
    float* const cold_lane = &s_cold[0][threadIdx.x];
    uint32_t& item = *(uint32_t*)(cold_lane + 11 * 256);
    typedef GeoVec<VR_GEO_IN_LDS(INTEG, EXACT, SPAN)> Geo;
    typename Geo::type eye = Geo::make(cold_lane + 14 * 256), dir = Geo::make(cold_lane + 17 * 256);   // current index-space ray (camera walk or shadow walk)
    // voxel-span / empty-run estimates only: 1/dir (NaN where dir == 0: fminf then ignores that axis)
    // and, per axis, (dir > 0 ? 1 - slack : slack) - eye, so that the ray leaves the shrunk voxel
    // whose corner is f at t = (f + offset) * inv = fma(f, inv, span_off) with span_off = offset * inv, which is what
    // is kept (its rounding moves the face by |eye| * 1e-7 voxels, far inside the slack)
    typename Geo::type inv = Geo::make(cold_lane + 20 * 256), span_off = Geo::make(cold_lane + 23 * 256);
    float    t = 0, t1 = 0;             // tFar, iRay.t1()
    LdsF3    wdir{cold_lane};           // data->rayDirection
    LdsF3    radiance{cold_lane + 3 * 256};   // data->radiance / result
    // data->transmission, shadow-walk transmission.  EXACT keeps them as the reference does.  FAST keeps -log2 of them
    // (optical depth in bits): attenuating by exp(-len mu_t) is then ONE fma in the step loop instead of a multiply,
    // a v_exp_f32 (quarter rate) and another multiply, the roulette's "T < 0.05" is "depth > log2(20)", its "T /= 0.25"
    // is "depth -= 2" (exact), a dead walk is +inf, and the value itself (trans()) is only needed in the event phase.
    constexpr float T_ONE = EXACT ? 1.0f : 0.0f;
    constexpr float DEPTH_ROULETTE = 4.321928094887362f;   // -log2(0.05)
    float    T = T_ONE, Tsh = T_ONE;
    auto trans = [](float x) -> float { return EXACT ? x : __builtin_amdgcn_exp2f(-x); };
    auto faint = [](float x) -> bool { return EXACT ? x < 0.05f : x > DEPTH_ROULETTE; };   // the roulette's condition
    // iRay.t0() of the camera ray is read when its shadow walk starts, which is where Tsh is first written, and
    // written again when the next camera ray starts, after Tsh was last read: the two share a register
    float&   t0 = Tsh;
    float    pdf = 1.0f;                // data->rayPDF (only read with useImportanceSampling)
    int&     depth_remaining = *(int*)(cold_lane + 10 * 256);
    LdsF3    scatter_p{cold_lane + 6 * 256};   // iRay(tFar) of the pending scatter event
    float&   scatter_len = cold_lane[9 * 256]; // its pathLength (ST_ABSORBED: mu_a of the absorbing step)
    // names the walk's stream (vr_device.h): pixel, and in one word sample (bits 0-19), bounce (20-27)
    // and the shadow-walk flag (bit 31)
    uint32_t& rng_pixel = *(uint32_t*)(cold_lane + 12 * 256);
    uint32_t& rng_ts = *(uint32_t*)(cold_lane + 13 * 256);
    uint32_t rng_step = 0;                                 // iterations of the walk so far
    uint32_t xs0 = 1, xs1 = 0, xs2 = 0, xs3 = 0;           // the stream's xoshiro128+ state
    float    geom_clock = 0.0f;                            // GEOM: what is left of the walk's exponential clock, in bits
    uint32_t geom_event = 0;                               // GEOM: the word that decides absorption / scattering at the collision
    // density at iRay(tFar), fetched at the END of the previous trip so that its latency hides
    // behind the loop overhead and the other waves of the SIMD
    float    cur_val = 0.0f;
    uint32_t cur_off = 0xffffffffu;     // its voxel offset (0xffffffff: outside the bbox)
    float    t_span = 0.0f;             // while t < t_span the walk is provably still inside that voxel
    // one density lookup per completed step of a walk: the walk's step index is added when the walk
    // ends (temperature lookups are counted where they happen)
    unsigned long long n_lookups = 0;
    uint32_t n_paths = 0, n_scatter = 0, n_shadow = 0, n_limit = 0, n_queries = 0;
    int&     medium = *(int*)(cold_lane + (MESH ? 26 : 11) * 256);   // MESH: object whose medium the walk is in (else unused alias)
    unsigned long long w_trips = 0, w_trip_lanes = 0, w_phases = 0;   // wave-uniform divergence counters
    uint32_t n_loads = 0;                     // voxel loads this lane issued (the other lookups found the voxel in its register)
    uint32_t chunk_next = 0, chunk_end = 0;   // wave-uniform: the wave's own range of work items
    bool     chunk_dry = false;               // every strip's counter has run past its last item
    uint32_t strip_try = 0, strip_lo = 0, strip_w = 0;   // wave-uniform: the strip of tile columns the chunk came from

    // value at iRay(t) for the coming trip (nearest voxel by floor, background outside).  Results
    // are collected in locals and assigned once at the end: every state variable is then written on
    // every path, which keeps them all in registers.
    auto prefetch = [&]() {
        if (SPAN && t < t_span) return;                  // same voxel, nothing to do
        const f3 q = ray_at(eye, dir, t);
        const float fx = floorf(q.x), fy = floorf(q.y), fz = floorf(q.z);
        const int32_t i = (int32_t)fx - p.density.min[0], j = (int32_t)fy - p.density.min[1], k = (int32_t)fz - p.density.min[2];
        const bool in = (uint32_t)i < (uint32_t)p.density.dim[0] && (uint32_t)j < (uint32_t)p.density.dim[1] &&
                        (uint32_t)k < (uint32_t)p.density.dim[2];
        float    n_val = p.density.background, n_span = t;
        uint32_t n_off = 0xffffffffu;
        if (in) {
            n_off = brick_offset(p.density, (uint32_t)i, (uint32_t)j, (uint32_t)k);
            n_val = cur_val;
            if (n_off != cur_off) { n_val = p.density.bricks[n_off]; VR_COUNT_LOAD(n_loads); }
            if (SPAN || SKIP)
                // ray parameter at which the walk leaves the voxel shrunk by `slack` on every side (see empty_run)
            {
                const f3 so_ = span_off, in_ = inv;
                n_span = fminf(fminf(fmaf(fx, in_.x, so_.x), fmaf(fy, in_.y, so_.y)), fmaf(fz, in_.z, so_.z));
            }
        }
        cur_val = n_val; cur_off = n_off; t_span = n_span;
    };
    // a lower bound of the same for the cube of `radius` voxels around the current voxel.  Per axis the cube's far face
    // lies radius * |1 / dir| beyond the voxel's, so the ray stays inside until at least t_span + radius * min |1 / dir|
    // (the exact minimum over the axes ends at most one voxel crossing later; it cost the block 20 instructions and
    // five more LDS reads on every trip: config 2 A/B in DESIGN.md 6b).  A shorter run never changes a result.
    auto cube_span = [&](float radius) -> float {
        const f3 in_ = inv;
        return fmaf(radius, fminf(fminf(fabsf(in_.x), fabsf(in_.y)), fabsf(in_.z)), t_span);
    };
    // a walk's stream: the Philox block of (pixel, sample, tag) is the generator's state
    auto seed_stream = [&]() {
        philox4x32_10(rng_pixel, rng_ts & 0xfffffu, ((rng_ts >> 20) & 0xffu) | (rng_ts & 0x80000000u), 0u, k0, k1, xs0, xs1, xs2, xs3);
        if ((xs0 | xs1 | xs2 | xs3) == 0u) xs0 = 1u;   // the one state xoshiro cannot leave
    };
    auto set_ray = [&](f3 e, f3 d, float a, float b) {
        eye = e; dir = d; t = a; t1 = b;
        const float slack = 0.01f, nan = __builtin_nanf("");
        inv = mk3(d.x != 0 ? __builtin_amdgcn_rcpf(d.x) : nan, d.y != 0 ? __builtin_amdgcn_rcpf(d.y) : nan,
                  d.z != 0 ? __builtin_amdgcn_rcpf(d.z) : nan);
        {
            const f3 in_ = inv;
            span_off = mk3(((d.x > 0 ? 1.0f - slack : slack) - e.x) * in_.x, ((d.y > 0 ? 1.0f - slack : slack) - e.y) * in_.y,
                           ((d.z > 0 ? 1.0f - slack : slack) - e.z) * in_.z);
        }
        t_span = a;   // force a lookup
        cur_off = 0xffffffffu;
        prefetch();
    };

    for (;;) {
        // =========================== event phase ===========================
        w_phases++;
#if VR_EVENT_PRIO >= 0
        if (SPAN) __builtin_amdgcn_s_setprio(VR_EVENT_PRIO);
#endif
        // (-1) FLAT iterations: real collisions of the camera walk, left undecided by the step loop.  The lane kept the event
        // word in cur_off and the step's length in t_span (both dead for a walk that has collided: its next ray resets them).
        if (FLAT && __any(st == ST_COLLIDED)) {
            if (st == ST_COLLIDED) {
                const float sigma = p.light_mult * cur_val;
                const float pA = sigma * fast_pa24, pS = sigma * fast_ps24;
                const float pNull = fmaxf(16777216.0f - pA - pS, 0.0f);
                const float sample = (float)(cur_off >> 8);
                if (sample < pNull + pA) {
                    // absorption: the medium colour is added below                          :127-131 (delta :101-113)
                    if (INTEG == INTEG_DELTA && TEMP) n_lookups++;               // accEmission.getValue, result unused (delta :103)
                    scatter_len = sigma * p.sigma_a;
                    st = ST_ABSORBED;
                } else {
                    // scattering                                                            :133-191 (delta :115-158)
                    depth_remaining--;                                                        // :136
                    n_scatter++;
                    scatter_p = ray_at(eye, dir, t);                                          // :162 (delta :130)
                    scatter_len = t_span;
                    st = INTEG == INTEG_DELTA ? ST_SHADOW_DONE : ST_SCATTER;
                }
            }
        }
        // (0) walks that escaped or were absorbed add their constant colour
        if (__any(st >= ST_ESCAPED)) {
            if (st >= ST_ESCAPED) {
                const TrackCold* c = opaque(p.cold);
                if (st == ST_ESCAPED) {
                    // survivor: radiance + background * transmission            IntegratorRatioTracking.cpp:208 (delta :175)
                    const f3 bg = mk3(c->background[0], c->background[1], c->background[2]);
                    radiance = add3(radiance, INTEG == INTEG_RATIO ? scale3(bg, trans(T)) : bg);
                } else {
                    const f3 medium = mk3(c->medium[0], c->medium[1], c->medium[2]);
                    if (INTEG == INTEG_RATIO) {                                   // :127-131
                        radiance = add3(radiance, scale3(medium, trans(T)));
                    } else {                                                      // delta :101-113
                        if (TEMP)
                            radiance = add3(radiance, scale3(mk3(c->emission[0], c->emission[1], c->emission[2]), scatter_len));
                        radiance = add3(radiance, medium);
                    }
                }
                st = ST_FINISHED;
            }
        }
        // (1) scatter events of the ratio tracker start their shadow walk        IntegratorRatioTracking.cpp:213-234
        if (INTEG == INTEG_RATIO && __any(st == ST_SCATTER)) {
            if (st == ST_SCATTER) {
                const TrackCold* c = opaque(p.cold);
                const f3 light = mk3(c->light[0], c->light[1], c->light[2]);
                n_shadow++;
                const f3 so = ray_at(eye, dir, t0);                               // :226 (segment START)
                n_lookups += rng_step;
                rng_ts |= 0x80000000u; rng_step = 0;
                float c0 = 1e-5f, c1 = VR_FLT_MAX;
                int  seg = SEG_FOUND;
                bool seg_empty = false;
                const bool in_box = ray_clip<EXACT>(so, light, c0, c1, c->bmin, c->bmax1);
                if (MESH && in_box) {
                    int m2 = medium;
                    seg = mesh_clip(c->mesh, index_to_world(c->mat, c->vec, so), mat_mult(c->mat, light), 1.0f, medium, m2, c0, c1, seg_empty, n_queries);
                }
                if (!in_box || (MESH && (seg == SEG_BLACK || seg_empty))) {       // :229-232
                    Tsh = (MESH && in_box && seg == SEG_BLACK) ? (EXACT ? 0.0f : __builtin_inff()) : T_ONE;     // an opaque surface between the point and the light
                    st = ST_SHADOW_DONE;
                } else {
                    shadow = true; Tsh = T_ONE; st = ST_WALK;
                    if (!(VR_ONE_STREAM_EXP && !EXACT)) seed_stream();
                    set_ray(so, light, c0, c1);
                }
            }
        }
        // (2) finished shadow walks (and every scatter of the delta tracker): light, new direction
        if (__any(st == ST_SHADOW_DONE)) {
            if (st == ST_SHADOW_DONE) {
                const TrackCold* c = opaque(p.cold);
                if (INTEG == INTEG_RATIO) {
                    rng_ts &= 0x7fffffffu;    // back on the camera walk
                    const f3 light = mk3(c->light[0], c->light[1], c->light[2]);
                    const float cos_theta = dot3(wdir, light);                            // :144
                    const float hg = hg_phase<EXACT>(c->g, cos_theta);                    // :145
                    f3 nee = scale3(mk3(c->light_color[0], c->light_color[1], c->light_color[2]), EXACT ? T * Tsh : trans(T + Tsh)); // :147-152
                    nee = scale3(nee, scatter_len);
                    nee = scale3(nee, hg);
                    radiance = add3(radiance, nee);
                }
                // the next bounce's stream starts with the direction sample
                n_lookups += rng_step;
                rng_ts += 1u << 20; rng_step = 0;
                if (!(VR_ONE_STREAM_EXP && !EXACT)) seed_stream();
                uint32_t h0, h1;
                xoshiro128p_next2(xs0, xs1, xs2, xs3, h0, h1);
                const float xi1 = in_range(h0, 0.0f, 0.9999f);
                const float xi2 = in_range(h1, 0.0f, 0.9999f);
                const f3 scattered = hg_sample<EXACT>(c->g, wdir, xi1, xi2);              // :156-159
                if (IS) {
                    // path pdf *= phase function of the sampled direction                 :169-176 (delta :137-144)
                    float cos_pdf;
                    if (INTEG == INTEG_RATIO) {
                        cos_pdf = dot3(scattered, wdir);
                    } else {
                        // the delta tracker takes the angle to the normalised PRIMARY direction (:22, :139)
                        const uint32_t x = item % c->cam.width, q = item / c->cam.width;
                        const uint32_t r = q % c->cam.nrows, s = c->cam.s_first + q / c->cam.nrows;
                        const f3 primary = primary_direction<EXACT>(c->cam, c->cam.x0 + x, c->cam.y0 + c->cam.row_first + r * c->cam.row_stride, s);
                        cos_pdf = dot3(M::normalize_(primary), wdir);
                    }
                    pdf *= hg_phase<EXACT>(c->g, cos_pdf);
                }
                wdir = scattered;                                                         // :178
                shadow = false;
                float c0 = 1e-5f, c1 = VR_FLT_MAX;
                int  seg = SEG_FOUND;
                bool seg_empty = false;
                const bool in_box = ray_clip<EXACT>(scatter_p, scattered, c0, c1, c->bmin, c->bmax1);
                if (MESH && in_box)
                    seg = mesh_clip(c->mesh, index_to_world(c->mat, c->vec, scatter_p), mat_mult(c->mat, scattered), 1.0f, medium, medium, c0, c1,
                                    seg_empty, n_queries);
                if (!in_box || (MESH && (seg == SEG_BLACK || seg_empty))) {               // :182-187 "scattering failed"
                    st = ST_FINISHED;
                } else if (depth_remaining <= 0) {
                    // `while (... && depthRemaining > 0)`: out of scatter budget, the walk ends as
                    // escaped                                                            :55, :208 (delta :54, :175)
                    const f3 bg = mk3(c->background[0], c->background[1], c->background[2]);
                    radiance = add3(radiance, INTEG == INTEG_RATIO ? scale3(bg, trans(T)) : bg);
                    st = ST_FINISHED;
                } else {
                    t0 = c0; st = ST_WALK;                                                // :179, :189
                    if (GEOM) {
                        uint32_t w_path;
                        xoshiro128p_next2(xs0, xs1, xs2, xs3, w_path, geom_event);
                        geom_clock = -__builtin_amdgcn_logf(fmaf((float)(w_path >> 8), (1.0f - 0.00001f) / 16777216.0f, 0.00001f));
                    }
                    set_ray(scatter_p, scattered, c0, c1);
                }
            }
        }
        // (3) finished paths store their sample
        if (st == ST_FINISHED) {
            n_lookups += rng_step; rng_step = 0;
            if (IS) {
                // color += rayResults[i] / std::max(0.1f, rayPDFs[i])                     BaseIntegrator.cpp:78
                const float den = (0.1f < pdf) ? pdf : 0.1f;
                const f3 r = radiance;
                radiance = mk3(r.x / den, r.y / den, r.z / den);
            }
            store_sample(p.work.samples, item, radiance);
            st = ST_NEED;
        }
        // (4) refill: idle lanes take the next items of the wave's chunk; the wave takes a new chunk
        // from its strip's counter when its own is used up.  (One atomic per refill on a single
        // address serialises at ~11 ns each in L2 and was the bound of every short-path workload.)
        for (;;) {
            const bool               want = st == ST_NEED;
            const unsigned long long need = __ballot(want);
            if (need == 0ull) break;
            if (chunk_next == chunk_end) {
                if (chunk_dry) {
                    if (want) st = ST_EXHAUSTED;
                    break;
                }
                // one vertical strip of tile columns per XCD (workgroups go round-robin over the XCDs, each
                // with an L2 of its own); a wave whose strip is used up helps with the next one
                const TrackCold* c = opaque(p.cold);
                const uint32_t tiles_x = (c->cam.width + (64u >> c->tile_h_log) - 1u) >> (6u - c->tile_h_log);
                const uint32_t tiles_y = (c->cam.nrows + (1u << c->tile_h_log) - 1u) >> c->tile_h_log;
                bool got = false;
                while (strip_try < VR_TRACK_STRIPS) {
                    const uint32_t strip = ((blockIdx.x & 7u) + strip_try) % VR_TRACK_STRIPS;
                    strip_lo = tiles_x * strip / VR_TRACK_STRIPS;
                    strip_w = tiles_x * (strip + 1u) / VR_TRACK_STRIPS - strip_lo;
                    const uint32_t strip_items = strip_w * tiles_y * c->cam.s_count * 64u;
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(p.work.counter + strip * 8u, p.work.chunk);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (base < strip_items) {
                        chunk_next = base;
                        chunk_end = base + p.work.chunk < strip_items ? base + p.work.chunk : strip_items;
                        got = true;
                        break;
                    }
                    strip_try++;
                }
                if (!got) { chunk_dry = true; continue; }
            }
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            const uint32_t avail = chunk_end - chunk_next, asked = (uint32_t)__popcll(need);
            const uint32_t first = chunk_next;
            chunk_next += asked < avail ? asked : avail;
            if (want && rank < avail) {
                const uint32_t mine = first + rank;
                {
                    const TrackCold* c = opaque(p.cold);
                    // Work order.  Items go out as tiles of 64 pixels (8x8), all sub-samples of a tile one after the other, two tile-samples
                    // to a chunk: their primary rays cross the same voxels, and every lookup misses L2 otherwise
                    // (the voxel-sized-step kernels were bound by L2 misses at a 42 % hit rate; against row segments
                    // handed out sub-sample by sub-sample: config 3 textbook +12.6 %, config 4 +21 %, config 2
                    // textbook +6 %, configs 2 / 3 +1.6 % / +2.2 %).  Within a strip the tile rows go from the middle
                    // of the frame outwards (middle, +1, -1, +2, ...): a long path takes 20-30 ms of wall time and
                    // the launch cannot end before its last path does; the medium is usually framed, so handing the
                    // edge rows out last leaves shorter paths for the end.
                    const uint32_t th_log = c->tile_h_log, tw_log = 6u - th_log;
                    const uint32_t tile = mine >> 6, l = mine & 63u;
                    const uint32_t sl = tile % c->cam.s_count;
                    const uint32_t q = tile / c->cam.s_count;
                    const uint32_t tx = strip_lo + q % strip_w;
                    const uint32_t k = q / strip_w;
                    const uint32_t tiles_y = (c->cam.nrows + (1u << th_log) - 1u) >> th_log;
                    const uint32_t ty = (k & 1u) ? tiles_y / 2 - (k + 1) / 2 : tiles_y / 2 + k / 2;
                    const uint32_t x = (tx << tw_log) + (l & ((1u << tw_log) - 1u)), r = (ty << th_log) + (l >> tw_log);
                    if (x >= c->cam.width || r >= c->cam.nrows) continue;   // padding of the last tile column / row
                    item = (sl * c->cam.nrows + r) * c->cam.width + x;
                    const uint32_t s = c->cam.s_first + sl;
                    const uint32_t ci = c->cam.x0 + x;
                    const uint32_t cj = c->cam.y0 + c->cam.row_first + r * c->cam.row_stride;
                    n_paths++;
                    // castRay prologue                      IntegratorRatioTracking.cpp:13-53
                    wdir = primary_direction<EXACT>(c->cam, ci, cj, s);
                    const f3 weye = mk3(c->cam.pos[0], c->cam.pos[1], c->cam.pos[2]);
                    // wRay.worldToIndexF(grid)               Ray.h:257-267, NanoVDB.h:2262-2270
                    const f3 ieye = mat_mult(c->inv_mat, mk3(weye.x - c->vec[0], weye.y - c->vec[1], weye.z - c->vec[2]));
                    const f3 idir = mat_mult(c->inv_mat, wdir);
                    const float length = M::sqrt_(idir.x * idir.x + idir.y * idir.y + idir.z * idir.z);
                    const float inv_length = M::rcp_(length);
                    const f3 ndir = mk3(inv_length * idir.x, inv_length * idir.y, inv_length * idir.z);
                    float c0 = length * 1e-5f, c1 = VR_FLT_MAX;
                    int  seg = SEG_FOUND;
                    bool seg_empty = false;
                    const bool in_box = ray_clip<EXACT>(ieye, ndir, c0, c1, c->bmin, c->bmax1);
                    if (MESH && in_box) {
                        medium = -1;
                        seg = mesh_clip(c->mesh, weye, wdir, length, -1, medium, c0, c1, seg_empty, n_queries);
                    }
                    if (MESH && in_box && seg == SEG_BLACK) {
                        store_sample(p.work.samples, item, mk3(0, 0, 0));                 // an opaque surface
                    } else if (!in_box || c->max_depth == 0 || (MESH && (seg == SEG_NONE || seg_empty))) {
                        // missed the grid (:37-39) or the mesh, or a zero scatter budget: the loop never runs (:55, :208)
                        store_sample(p.work.samples, item, mk3(c->background[0], c->background[1], c->background[2]));
                    } else {
                        st = ST_WALK; shadow = false;
                        t0 = c0;
                        radiance = mk3(0, 0, 0);
                        T = T_ONE;
                        depth_remaining = (int)c->max_depth;
                        rng_pixel = (cj << 16) | ci; rng_ts = s; rng_step = 0;
                        seed_stream();
                        pdf = 1.0f;                                                       // :14
                        if (GEOM) {
                            uint32_t w_path;
                            xoshiro128p_next2(xs0, xs1, xs2, xs3, w_path, geom_event);
                            geom_clock = -__builtin_amdgcn_logf(fmaf((float)(w_path >> 8), (1.0f - 0.00001f) / 16777216.0f, 0.00001f));
                        }
                        set_ray(ieye, ndir, c0, c1);
                    }
                }
            }
        }
        if (!__any(st == ST_WALK)) break;   // every lane exhausted (events were all served above)

        // =========================== step loop ===========================
        // One trip = up to VR_TRIP_ITERS iterations of the reference's while loop.  Few lanes are in empty space at
        // any time but nearly every wave has one, so the empty-space code runs once per trip, after
        // the iterations: a lane that finds sigma <= 0 sits out the rest of the trip and takes its
        // run there.
        if (st == ST_WALK && rng_step > p.work.step_limit) { n_limit++; st = ST_FINISHED; }   // safety cap, once per round
#if VR_EVENT_PRIO >= 0
        if (SPAN) __builtin_amdgcn_s_setprio(VR_STEP_PRIO);
#endif
#pragma unroll 1
        for (int it = 0; it < p.round_max; ++it) {
          if constexpr (GEOM) {
            // one RUN per trip: every step that starts in the current voxel (dense: stepSizeMin each, the clock decides where
            // the real collision falls) or in the cube of empty voxels around it (10 * stepSizeMin each, no collision)
            if (st == ST_WALK) {
                const float sigma = p.light_mult * cur_val;
                const bool  dense = sigma > 0.0f;
                const float step = dense ? p.t_min : p.t_min * 10;                            // :62-72, every free path clamped
                // pNull as the reference rounds it (:84-86), the step's real-collision probability 1 - pNull (exact in float)
                // and its cost on the clock, -log2(pNull): by series where 1 - pNull is small (v_log_f32 near 1 is not)
                const float pA = M::div_(sigma * p.sigma_a, p.sigma_maj), pS = M::div_(sigma * p.sigma_s, p.sigma_maj);
                const float pNull = fmaxf(1.0f - pA - pS, 0.0f);
                const float pReal = 1.0f - pNull;
                const float series = pReal * fmaf(pReal, fmaf(pReal, fmaf(pReal, 0.25f, 1.0f / 3.0f), 0.5f), 1.0f) * 1.4426950408889634f;
                float cost = pReal < 0.05f ? series : -__builtin_amdgcn_logf(pNull);
                if (!dense) cost = 0.0f;
                float t_safe = t_span;
                if (!dense && cur_off != 0xffffffffu && cur_val < -1.5f) t_safe = cube_span(-1.0f - cur_val);
                t_safe = t_safe < t1 ? t_safe : t1;
                // empty_run's arithmetic (vr_device.h): kf steps that start AND end inside the box, each adding exactly qu
                const float    qu = (t + step) - t;
                const uint32_t tb = __float_as_uint(t);
                const float    half_ulp = __uint_as_float(tb & 0x7f800000u) * 5.9604644775390625e-08f;
                const bool     regular = qu > 0.0f && fabsf(qu - step) != half_ulp;
                const float    t_last = __uint_as_float(tb | 0x7fffffu);
                const float    t_lim = t_safe < t_last ? t_safe : t_last;
                float kf = floorf((t_lim - t) * __builtin_amdgcn_rcpf(qu) * 0.999999f) - 1.0f;
                if (!(regular && kf >= 1.0f) || cur_off == 0xffffffffu) kf = 0.0f;
                // the step (1-based) at which the clock runs out: the smallest j with j * cost > clock
                const float jf = fmaxf(floorf(geom_clock * __builtin_amdgcn_rcpf(cost)), 0.0f) + 1.0f;   // cost 0: +inf, cost +inf: 1
                bool collided = false;
                if (jf <= kf) {
                    t = fmaf(jf, qu, t);
                    rng_step += (uint32_t)jf;
                    collided = true;
                } else {
                    t = fmaf(kf, qu, t);
                    t += step;                                                                // an ordinary step: may leave the box
                    rng_step += (uint32_t)kf + 1u;
                    if (t > t1) st = ST_ESCAPED;                                              // :77-79
                    else if (jf == kf + 1.0f) collided = true;
                    else geom_clock = fmaf(-(kf + 1.0f), cost, geom_clock);
                }
                if (collided) {
                    // absorption or scattering, given a real collision: sample uniform in [pNull, 1)     :93-115
                    if ((float)(geom_event >> 8) * (1.0f / 16777216.0f) * pReal < pA) {
                        if (TEMP) n_lookups++;                                                // accEmission.getValue (:103)
                        scatter_len = sigma * p.sigma_a;
                        st = ST_ABSORBED;
                    } else {
                        depth_remaining--;                                                    // :118
                        n_scatter++;
                        scatter_p = ray_at(eye, dir, t);                                      // :130
                        scatter_len = step;
                        st = ST_SHADOW_DONE;
                    }
                }
            }
            if (st == ST_WALK) prefetch();
            const int n_walk = __popcll(__ballot(st == ST_WALK));
            w_trips++; w_trip_lanes += (unsigned)n_walk;
            const int n_event = __popcll(__ballot(st >= ST_SCATTER));
            if (n_event >= p.round_events || n_walk == 0) break;
          } else {
#if VR_SERVICE_PRIO >= 0
            if (SPAN) __builtin_amdgcn_s_setprio(VR_STEP_PRIO);
#endif
            bool empty = false;     // this lane's current iteration is an empty-space step, still to be taken
            bool parked = false;    // this lane has left its voxel and waits for the lookup at the end of the trip
#pragma unroll
            for (int half = 0; half < VR_TRIP_ITERS; ++half) {   // (`half`: the trips once had two)
                if (st != ST_WALK || empty || parked) continue;
                const bool in_shadow = (INTEG == INTEG_RATIO) && shadow;
                if constexpr (FLAT) {
                    const float mult = in_shadow ? p.shadow_mult : p.light_mult;
                    const float sigma = mult * cur_val;                                       // :71 / :238
                    const bool  roulette = INTEG == INTEG_RATIO && !in_shadow && faint(T);
                    if (!roulette && !(sigma > 0.0f)) { empty = true; continue; }
                    uint32_t w_path, w_event;
                    xoshiro128p_next2(xs0, xs1, xs2, xs3, w_path, w_event);
                    if (roulette) {                                                           // :57-69 (rare)
                        if (roulette16(w_path, w_event) < 0.75f) { st = ST_FINISHED; continue; }
                        T -= 2.0f;
                        if (!(sigma > 0.0f)) { empty = true; continue; }
                    }
                    const float tmin = in_shadow ? p.t_min * 5 : p.t_min;                     // :216
                    const float u = fmaf((float)(w_path >> 8), (1.0f - 0.00001f) / 16777216.0f, 0.00001f);
                    // (what follows derives c, the null threshold and the depth per length from sigma in every iteration; keeping the
                    // three per voxel, worked out by the lookup, was measured: 12 bytes of spills and 0.5-3 % slower)
                    const float c = in_shadow ? fast_c_ln * __builtin_amdgcn_rcpf(sigma * fast_sig_t) : fast_c_cam;
                    const float len = __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(u) * c, tmin, p.t_max);   // :80-82 / :247-249
                    t += len;                                                                 // :88 / :255
                    rng_step += 1;
                    const bool exited = t > t1;                                               // :91 / :258
                    if (half + 1 < VR_TRIP_ITERS) {
                        if (SPAN) parked = !exited && !(t < t_span);
                        else if (!exited) prefetch();
                    }
                    // the step's optical depth in bits (nothing for the step that leaves the segment)  :103-105 / :266-269
                    const float att = (INTEG == INTEG_RATIO && !exited) ? len * (sigma * fast_sig_t2) : 0.0f;
                    if (IS && !in_shadow && !exited) pdf *= (sigma * fast_sig_t) * __builtin_amdgcn_exp2f(-(len * (sigma * fast_sig_t2)));
                    // camera walk: null collision?                                           :98-100, :108
                    const float pNull = fmaxf(16777216.0f - sigma * fast_pa24 - sigma * fast_ps24, 0.0f);
                    const bool  real = !in_shadow && !exited && !((float)(w_event >> 8) < pNull);
                    if (INTEG == INTEG_RATIO) {
                        const float ts = Tsh + att;
                        T = in_shadow ? T : T + att;
                        Tsh = in_shadow ? ts : Tsh;
                    }
                    st = exited ? (in_shadow ? (uint32_t)ST_SHADOW_DONE : (uint32_t)ST_ESCAPED) : (real ? (uint32_t)ST_COLLIDED : st);
                    cur_off = real ? w_event : cur_off;      // the event phase decides absorption / scattering from it
                    t_span = real ? len : t_span;            // ... and needs the step's length (scatter_len)
                    if (INTEG == INTEG_RATIO && in_shadow && !exited && Tsh > DEPTH_ROULETTE) {   // :272-281 (rare)
                        if (in_range(w_event, 0.0f, 1.0f) < 0.75f) { Tsh = __builtin_inff(); st = ST_SHADOW_DONE; }
                        else Tsh -= 2.0f;
                    }
                    continue;
                }

                // sigma = multiplier * acc.getValue(Coord::Floor(iRay(tFar)))                :71 / :238
                const float mult = in_shadow ? p.shadow_mult : p.light_mult;
                const float sigma = mult * cur_val;

                const bool roulette = INTEG == INTEG_RATIO && !in_shadow && faint(T);
                if (!roulette && !(sigma > 0.0f)) { empty = true; continue; }
                // the iteration's two words
                uint32_t w_path, w_event;
                xoshiro128p_next2(xs0, xs1, xs2, xs3, w_path, w_event);
                if (roulette) {
                    // Russian roulette on the camera walk     IntegratorRatioTracking.cpp:57-69
                    // (a path it ends does not count the lookup above, as in the reference)
                    const float q = 0.75f;
                    if (EXACT) {
                        if (roulette16(w_path, w_event) < q) T = 0.0f;
                        else T /= 1.0f - q;
                        if (T <= 0.0f) { st = ST_FINISHED; continue; }
                    } else {
                        if (roulette16(w_path, w_event) < q) { st = ST_FINISHED; continue; }
                        T -= 2.0f;
                    }
                    if (!(sigma > 0.0f)) { empty = true; continue; }
                }

                // free path: camera walk over the global majorant, shadow walk over mu_t   :80-82 / :247-249
                const float tmin = in_shadow ? p.t_min * 5 : p.t_min;                         // :216
                float len;
                if (EXACT) {
                    const float mu_t = sigma * p.sigma_a + sigma * p.sigma_s;
                    const float denom = in_shadow ? mu_t : p.sigma_maj;
                    len = M::div_(-M::log_(in_range(w_path, 0.00001f, 1.0f)), denom);
                    len *= p.step_mult;
                    len = clamp_lo_hi(tmin, p.t_max, len);
                } else {
                    // same quantities with the constants folded: -ln 2 * stepSizeMultiplier / denominator
                    const float u = fmaf((float)(w_path >> 8), (1.0f - 0.00001f) / 16777216.0f, 0.00001f);
                    const float c = in_shadow ? fast_c_ln * __builtin_amdgcn_rcpf(sigma * fast_sig_t) : fast_c_cam;
                    len = __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(u) * c, tmin, p.t_max);
                }
                t += len;                                                                     // :88 / :255
                rng_step += 1;
                // the next iteration's voxel.  With sub-voxel steps (SPAN) few lanes need one,
                // yet nearly every wave has such a lane: it sits out the rest of the trip instead and
                // is served by the lookup at the end of the trip.
                if (half + 1 < VR_TRIP_ITERS && t <= t1) {
                    if (SPAN) parked = !(t < t_span);
                    else prefetch();
                }
                if (t > t1) {                                                                 // :91 / :258
                    st = in_shadow ? ST_SHADOW_DONE : ST_ESCAPED;                             // shadow walk returns its transmission
                    continue;
                }
                // attenuation over the step: exp(-len * mu_t)                                :103-105 / :266-269
                float att = 1.0f, mu_a = 0.0f;
                if (EXACT) {
                    mu_a = sigma * p.sigma_a;
                    const float mu_t = mu_a + sigma * p.sigma_s;
                    if (INTEG == INTEG_RATIO || IS) att = M::exp_(-(len)*mu_t);
                    // rayPDF *= mu_t * sampleAttenuation (a double there)                 :106 (delta :91)
                    if (IS && !in_shadow) pdf = (float)((double)pdf * ((double)mu_t * (double)att));
                } else {
                    // FAST: `att` is the step's optical depth in bits, len * mu_t * log2(e)
                    if (INTEG == INTEG_RATIO) att = len * (sigma * fast_sig_t2);
                    if (IS && !in_shadow) pdf *= (sigma * fast_sig_t) * __builtin_amdgcn_exp2f(-(len * (sigma * fast_sig_t2)));
                }
                if (in_shadow) {
                    const float q = 0.75f;
                    if (EXACT) {
                        Tsh *= att;
                        if (Tsh < 0.05f) {                                                    // :272-279
                            if (in_range(w_event, 0.0f, 1.0f) < q) Tsh = 0.0f;
                            else Tsh /= 1.0f - q;
                        }
                        if (Tsh <= 0.0f) st = ST_SHADOW_DONE;                                 // :281
                    } else {
                        Tsh += att;
                        if (Tsh > DEPTH_ROULETTE) {
                            if (in_range(w_event, 0.0f, 1.0f) < q) { Tsh = __builtin_inff(); st = ST_SHADOW_DONE; }
                            else Tsh -= 2.0f;
                        }
                    }
                } else {
                    // collision type: null / absorption / scattering                         :98-100, :108
                    bool is_null, is_absorb;
                    if (EXACT) {
                        const float mu_s = sigma * p.sigma_s;
                        const float pA = M::div_(mu_a, p.sigma_maj);
                        const float pS = M::div_(mu_s, p.sigma_maj);
                        const float pn = 1 - pA - pS;
                        const float pNull = (0 < pn) ? pn : 0;
                        const float sample = in_range(w_event, 0.0f, 1.0f);
                        is_null = sample < pNull;
                        is_absorb = sample < pNull + pA;
                    } else {
                        // the same comparisons scaled by 2^24 (exact), against the raw 24-bit draw
                        const float pA = sigma * fast_pa24, pS = sigma * fast_ps24;
                        const float pNull = fmaxf(16777216.0f - pA - pS, 0.0f);
                        const float sample = (float)(w_event >> 8);
                        is_null = sample < pNull;
                        is_absorb = sample < pNull + pA;
                    }
                    if (INTEG == INTEG_RATIO) T = EXACT ? T * att : T + att;
                    if (INTEG == INTEG_RATIO && TEMP) {                          // :110-122
                        const TrackCold* c = opaque(p.cold);
                        const float emission = grid_value_nocache(c->temperature, ray_at(eye, dir, t));
                        n_lookups++;
                        n_loads++;
                        if (emission > 0.0f) {
                            f3 e = scale3(mk3(c->emission[0], c->emission[1], c->emission[2]), trans(T) * (sigma * p.sigma_a));
                            e = scale3(e, emission);
                            e = scale3(e, len);
                            radiance = add3(radiance, e);
                        }
                    }
                    if (is_null) {
                        // null collision
                    } else if (is_absorb) {
                        // absorption: the medium colour is added in the event phase          :127-131 (delta :101-113)
                        if (INTEG == INTEG_DELTA && TEMP) n_lookups++;           // accEmission.getValue, result unused (delta :103)
                        scatter_len = sigma * p.sigma_a;
                        st = ST_ABSORBED;
                    } else {
                        // scattering                                                        :133-191 (delta :115-158)
                        depth_remaining--;                                                    // :136 (the `< 0` test never fires)
                        n_scatter++;
                        if (INTEG == INTEG_DELTA || !EXACT || T > 0.0f) {                     // :141 (FAST: a walk the roulette ended never gets here)
                            scatter_p = ray_at(eye, dir, t);                                  // :162 (delta :130)
                            scatter_len = len;
                            st = INTEG == INTEG_DELTA ? ST_SHADOW_DONE : ST_SCATTER;
                        }
                    }
                }
            }
#if VR_SERVICE_PRIO >= 0
            if (SPAN) __builtin_amdgcn_s_setprio(VR_SERVICE_PRIO);
#endif
            // empty-space steps: tFar += constant, no draw                                   :84-96 / :251-263 (delta :70-82)
            if (empty) {
                const bool  in_shadow = (INTEG == INTEG_RATIO) && shadow;
                const float tmin = in_shadow ? p.t_min * 5 : p.t_min;
                const float len = in_shadow ? tmin * 100 : tmin * 10;                         // :252 / :85
                uint32_t    krun = 1;
                // a run must not span iterations that would play roulette (T < 0.05 on the camera walk)
                const bool can_run = INTEG == INTEG_DELTA || in_shadow || !faint(T);
                if (SKIP && cur_off != 0xffffffffu && can_run) {
                    // the skip grid holds -(r + 1) where every voxel of the cube of radius r is empty
                    float t_safe = t_span;
                    if (cur_val < -1.5f) t_safe = cube_span(-1.0f - cur_val);
                    krun = empty_run(t, t_safe < t1 ? t_safe : t1, len);   // (no NaNs here: a compare + select, not fminf's canonicalisation)
                } else {
                    t += len;
                }
                rng_step += krun;
                if (t > t1) st = in_shadow ? ST_SHADOW_DONE : ST_ESCAPED;                     // :91 / :258
            }
            // the voxel of the next trip's first iteration
            if (st == ST_WALK) prefetch();
            // leave for the event phase once enough lanes wait for it, or none walks any more
            const int n_walk = __popcll(__ballot(st == ST_WALK));
#ifdef VR_DRAIN_EXP
            if (chunk_dry) { w_trips++; w_trip_lanes += (unsigned)n_walk; }   // experiment: the divergence counters see the drain only
#elif defined(VR_EMPTY_EXP)
#ifndef VR_EMPTY_EXP_D
#define VR_EMPTY_EXP_D 0
#endif
            // experiment: trips that ran the empty-space block, and the lanes in it whose voxel has a distance of at least VR_EMPTY_EXP_D
            if (__any(empty)) { w_trips++; w_trip_lanes += (unsigned)__popcll(__ballot(empty && !(cur_val > -(float)VR_EMPTY_EXP_D))); }
#elif defined(VR_SHADOW_EXP)
            // experiment: trips in which no walking lane is on a camera walk, and the camera-walking lanes over all trips
            { const unsigned long long cam = __ballot(st == ST_WALK && !shadow); if (cam == 0ull) w_trips++; w_trip_lanes += (unsigned)__popcll(cam); }
#elif defined(VR_PARK_EXP)
            if (__any(parked)) { w_trips++; w_trip_lanes += (unsigned)__popcll(__ballot(parked)); }  // experiment: trips with lanes that left their voxel early
#else
            w_trips++; w_trip_lanes += (unsigned)n_walk;
#endif
            const int n_event = __popcll(__ballot(st >= ST_SCATTER));
            if (n_event >= p.round_events || n_walk == 0) break;
          }
        }
    }

    // ---- counters: one atomic per wave and counter ----
    unsigned long long a = wave_sum(n_paths), b = wave_sum(n_lookups), c = wave_sum(n_scatter), d = wave_sum(n_shadow),
                       e = wave_sum(n_limit);
    const unsigned long long nq = MESH ? wave_sum(n_queries) : 0ull;
    const unsigned long long nl = wave_sum(n_loads);
    if (lane == 0) {
        if (MESH) atomicAdd(&p.work.counters->ray_queries, nq);
        atomicAdd(&p.work.counters->path_samples, a);
        atomicAdd(&p.work.counters->density_lookups, b);
        atomicAdd(&p.work.counters->scatter_events, c);
        atomicAdd(&p.work.counters->shadow_rays, d);
        if (e) atomicAdd(&p.work.counters->step_limit_hits, e);
        atomicAdd(&p.work.counters->wave_trips, w_trips);
        atomicAdd(&p.work.counters->wave_trip_lanes, w_trip_lanes);
        atomicAdd(&p.work.counters->event_phases, w_phases);
        atomicAdd(&p.work.counters->voxel_loads, nl);
    }
}

// =============================================================================================
// delta tracking over per-brick majorants (VR_MODE_FAST_LOCAL) — NOT the reference's estimator
// =============================================================================================
// north_star "DensityFields": a coarse majorant grid and DDA traversal with tight per-cell bounds.  The reference
// samples every free path against ONE majorant, max(density) * (sigma_a + sigma_s) (IntegratorDeltaTracking.cpp:44-47),
// so in a cloud whose density is mostly far below its maximum nearly every tentative collision is a null one.  Here
// the ray walks the grid's 8^3 bricks (3-D DDA) and samples its free path against the brick's own maximum: a free
// path that ends beyond the brick is cut at the brick's exit and drawn again there (the exponential is memoryless),
// an empty brick is crossed in one step, and the density is looked up AT the tentative collision.  That is textbook
// delta tracking — unbiased for any field — whereas the reference decides a collision with the density it looked up
// one free path EARLIER (:55, :74, :84-98) and clamps / scales its free paths (:63-65).  So this mode is offered only
// where the reference's own parameters make it a plain delta tracker (stepSizeMultiplier 1, clamps out of reach: the
// "textbook" regime of SURVEY 7(ii), BASELINE configs 3-textbook and 4), it is opt-in, and its parity is statistical:
// exp(-tau) on slabs, layered ones included, and agreement with the global-majorant kernel within the smear of one
// free path (tests/test_gpu_local_majorant.py).
struct DdaParams {
    const TrackCold* cold;
    WorkDev          work;
    GridDev          density;
    const float*     major;          // per majorant cell: max over its voxels
    int32_t          cell_log;       // a majorant cell is (1 << cell_log)^3 voxels: 3 = the 8^3 bricks, 4 or 5 = 2^3 or 4^3 bricks
    int32_t          nc[3];          // cells per axis
    float            sigma_a, sigma_s, light_mult;
    int32_t          round_max, round_events;
};
enum { DD_NEED = 0, DD_WALK = 1, DD_EXHAUSTED = 2, DD_SCATTER = 3, DD_ABSORBED = 4, DD_ESCAPED = 5, DD_BLACK = 6 };

// Waves per SIMD: the walk's state (ray, DDA, generator, counters) is some 75 registers.  At eight waves (64 VGPRs) the
// compiler spilled 96 bytes per lane INSIDE the step loop and every trip paid scratch traffic on top of its own load:
// six waves without spills are 1.65 x as fast (config 3 textbook 133 -> 81 ms, config 4 764 -> 465 ms); five 5 % slower
// than six, seven (72 VGPRs, spills again) slower than eight.
#ifndef VR_DDA_WAVES
#define VR_DDA_WAVES 6
#endif
// EXACT (VR_MODE_EXACT_LOCAL): the arithmetic the oracle's cast_ray_delta_local shares — IEEE divisions and square roots, the
// polynomial log / sin / cos of vr_math.h — so that the kernel can be compared with a CPU restatement bit for bit; four
// waves per SIMD.  The production build (FAST) folds constants and uses the hardware's one-ulp forms.
// LDSMAJ (north_star "majorant bricks staged in LDS"): when the majorant grid has at most VR_DDA_LDS_CELLS cells every
// block copies it into LDS once (16 KB; with the 8 KB of event state six blocks still fit a CU) and the DDA reads a cell's
// majorant from there instead of from L2 — the walk enters a new cell on two trips out of three and waits for that load.
#define VR_DDA_LDS_CELLS 4096
#ifndef VR_DDA_MESH_WAVES_LESS
#define VR_DDA_MESH_WAVES_LESS 2   // the mesh-bounded build gets this many waves per SIMD fewer: at four (128 VGPRs) closest_hit runs without the 192 bytes of
                                   // scratch it spilled at five — config 5's delta leg 1342 -> 1102 ms; three waves: 1181
#endif
// MESH (meshDeltaTracking in the LOCAL modes; no reference counterpart, like MESH in vr_track_kernel): every ray of the walk —
// the primary ray, every scattered ray — is cut to the mesh's entry/exit interval in the event code (mesh_clip), and the lane
// remembers the object whose medium it is in.  Four waves per SIMD: closest_hit needs registers beside the walk's state.
template <bool TEMP, bool EXACT = false, bool LDSMAJ = false, bool MESH = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(EXACT ? 4 : VR_DDA_WAVES - (MESH ? VR_DDA_MESH_WAVES_LESS : 0), EXACT ? 4 : VR_DDA_WAVES - (MESH ? VR_DDA_MESH_WAVES_LESS : 0))))
vr_delta_dda_kernel(const DdaParams p)
{
    typedef Math<EXACT> M;
    __shared__ float s_major[LDSMAJ ? VR_DDA_LDS_CELLS : 1];
    if (LDSMAJ) {
        const uint32_t ncells = (uint32_t)p.nc[0] * (uint32_t)p.nc[1] * (uint32_t)p.nc[2];
        for (uint32_t i = threadIdx.x; i < ncells; i += 256u) s_major[i] = p.major[i];
        __syncthreads();
    }
    auto major_at = [&](uint32_t idx) -> float { return LDSMAJ ? s_major[idx] : p.major[idx]; };
    const int      lane = threadIdx.x & 63;
    const uint32_t k0 = p.work.key0, k1 = p.work.key1;
    const float    sig_t = p.sigma_a + p.sigma_s;
    const float    mu_scale = p.light_mult * sig_t;                   // brick maximum -> majorant of mu_t in the brick

    uint32_t st = DD_NEED;
    f3       eye = mk3(0, 0, 0), dir = mk3(0, 0, 1);                  // the index-space ray of the walk
    float    t = 0, t1 = 0;
    float    tmx = 0, tmy = 0, tmz = 0, tdx = 0, tdy = 0, tdz = 0;    // DDA: parameter of the next brick face per axis, and per brick
    int32_t  bi = 0, bj = 0, bk = 0;                                  // current brick
    float    mu = 0;                                                  // majorant of the current brick
    uint32_t xs0 = 1, xs1 = 0, xs2 = 0, xs3 = 0;
    __shared__ float s_ev[MESH ? 9 : 8][256];                         // what only the event code touches
    float* const ev = &s_ev[0][threadIdx.x];
    int&      medium = *(int*)(ev + (MESH ? 8 : 7) * 256);            // MESH: object whose medium the walk is in (else an unused alias)
    uint32_t  n_queries = 0;
    LdsF3     wdir{ev};                                               // data->rayDirection
    uint32_t& item = *(uint32_t*)(ev + 3 * 256);
    uint32_t& rng_pixel = *(uint32_t*)(ev + 4 * 256);
    uint32_t& rng_ts = *(uint32_t*)(ev + 5 * 256);
    int&      depth_remaining = *(int*)(ev + 6 * 256);
    float&    absorbed_mu_a = ev[7 * 256];
    uint32_t  steps = 0;
    unsigned long long n_lookups = 0, w_trips = 0, w_trip_lanes = 0, w_phases = 0;
    uint32_t n_paths = 0, n_scatter = 0, n_limit = 0;
    uint32_t chunk_next = 0, chunk_end = 0, strip_try = 0, strip_lo = 0, strip_w = 0;
    bool     chunk_dry = false;

    auto seed_stream = [&]() {
        philox4x32_10(rng_pixel, rng_ts & 0xfffffu, (rng_ts >> 20) & 0xffu, 0u, k0, k1, xs0, xs1, xs2, xs3);
        if ((xs0 | xs1 | xs2 | xs3) == 0u) xs0 = 1u;
    };
    // the DDA's state for cell (i, j, k) of the current ray: parameters of its far faces, per-cell increments, majorant
    const int32_t cl = p.cell_log;
    const float   cs = (float)(1 << cl);
    auto enter_brick = [&](int32_t i, int32_t j, int32_t k) {
        bi = i; bj = j; bk = k;
        const f3 e = eye, d = dir;
        const float inf = __builtin_huge_valf();
        const float ix = d.x != 0 ? M::rcp_(d.x) : inf, iy = d.y != 0 ? M::rcp_(d.y) : inf, iz = d.z != 0 ? M::rcp_(d.z) : inf;
        tdx = cs * fabsf(ix); tdy = cs * fabsf(iy); tdz = cs * fabsf(iz);
        tmx = d.x != 0 ? ((float)(p.density.min[0] + ((i + (d.x > 0 ? 1 : 0)) << cl)) - e.x) * ix : inf;
        tmy = d.y != 0 ? ((float)(p.density.min[1] + ((j + (d.y > 0 ? 1 : 0)) << cl)) - e.y) * iy : inf;
        tmz = d.z != 0 ? ((float)(p.density.min[2] + ((k + (d.z > 0 ? 1 : 0)) << cl)) - e.z) * iz : inf;
        mu = major_at(((uint32_t)i * (uint32_t)p.nc[1] + (uint32_t)j) * (uint32_t)p.nc[2] + (uint32_t)k) * mu_scale;
    };
    // start a walk along (e, d) over [a, b]: the cell of its first point
    auto set_ray = [&](f3 e, f3 d, float a, float b) {
        eye = e; dir = d; t = a; t1 = b;
        const f3 q = ray_at(e, d, a);
        const int32_t nb0 = p.nc[0], nb1 = p.nc[1], nb2 = p.nc[2];
        int32_t i = (floor_to_int(q.x) - p.density.min[0]) >> cl, j = (floor_to_int(q.y) - p.density.min[1]) >> cl,
                k = (floor_to_int(q.z) - p.density.min[2]) >> cl;
        i = i < 0 ? 0 : (i >= nb0 ? nb0 - 1 : i);     // the first point lies ON the bbox: rounding may put it a voxel outside
        j = j < 0 ? 0 : (j >= nb1 ? nb1 - 1 : j);
        k = k < 0 ? 0 : (k >= nb2 ? nb2 - 1 : k);
        enter_brick(i, j, k);
    };

    for (;;) {
        // =========================== event phase ===========================
        w_phases++;
        if (__any(st >= DD_SCATTER)) {
            const TrackCold* c = opaque(p.cold);
            if (st == DD_SCATTER) {
                // IntegratorDeltaTracking.cpp:115-158: a new direction from the phase function, the walk goes on from
                // the collision point
                // (one stream per PATH here: the walk after a scatter goes on drawing from the generator the refill
                // seeded.  The other kernels key a stream per bounce so that the oracle can restate them walk by walk;
                // this estimator has no oracle to follow, and Philox at every scatter was a tenth of its event phases.)
                n_scatter++;
                depth_remaining--;
                uint32_t h0, h1;
                xoshiro128p_next2(xs0, xs1, xs2, xs3, h0, h1);
                const f3 scattered = hg_sample<EXACT>(c->g, wdir, in_range(h0, 0.0f, 0.9999f), in_range(h1, 0.0f, 0.9999f));   // :124-127
                const f3 point = ray_at(eye, dir, t);                                         // :130
                wdir = scattered;                                                             // :146
                float c0 = 1e-5f, c1 = VR_FLT_MAX;
                int  seg = SEG_FOUND;
                bool seg_empty = false;
                const bool in_box = ray_clip<EXACT>(point, scattered, c0, c1, c->bmin, c->bmax1);
                if (MESH && in_box)
                    seg = mesh_clip(c->mesh, index_to_world(c->mat, c->vec, point), mat_mult(c->mat, scattered), 1.0f, medium, medium, c0, c1,
                                    seg_empty, n_queries);
                if (!in_box || (MESH && (seg == SEG_BLACK || seg_empty))) st = DD_BLACK;      // :150-155 "scattering failed"
                else if (depth_remaining <= 0) st = DD_ESCAPED;                               // :54, :175
                else { st = DD_WALK; set_ray(point, scattered, c0, c1); }
            }
            if (st >= DD_ABSORBED) {
                f3 colour = mk3(0, 0, 0);
                if (st == DD_ESCAPED) colour = mk3(c->background[0], c->background[1], c->background[2]);   // :175
                if (st == DD_ABSORBED) {                                                      // :101-113
                    colour = mk3(c->medium[0], c->medium[1], c->medium[2]);
                    if (TEMP) colour = add3(scale3(mk3(c->emission[0], c->emission[1], c->emission[2]), absorbed_mu_a), colour);
                }
                n_lookups += steps; steps = 0;
                store_sample(p.work.samples, item, colour);
                st = DD_NEED;
            }
        }
        // refill (the tracking kernel's work order: 8x8-pixel tiles, sub-samples of a tile together, one strip of tile
        // columns per XCD, tile rows from the middle of the frame outwards)
        for (;;) {
            const bool               want = st == DD_NEED;
            const unsigned long long need = __ballot(want);
            if (need == 0ull) break;
            const TrackCold* c = opaque(p.cold);
            if (chunk_next == chunk_end) {
                if (chunk_dry) { if (want) st = DD_EXHAUSTED; break; }
                const uint32_t tiles_x = (c->cam.width + (64u >> c->tile_h_log) - 1u) >> (6u - c->tile_h_log);
                const uint32_t tiles_y = (c->cam.nrows + (1u << c->tile_h_log) - 1u) >> c->tile_h_log;
                bool got = false;
                while (strip_try < VR_TRACK_STRIPS) {
                    const uint32_t strip = ((blockIdx.x & 7u) + strip_try) % VR_TRACK_STRIPS;
                    strip_lo = tiles_x * strip / VR_TRACK_STRIPS;
                    strip_w = tiles_x * (strip + 1u) / VR_TRACK_STRIPS - strip_lo;
                    const uint32_t strip_items = strip_w * tiles_y * c->cam.s_count * 64u;
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(p.work.counter + strip * 8u, p.work.chunk);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (base < strip_items) {
                        chunk_next = base;
                        chunk_end = base + p.work.chunk < strip_items ? base + p.work.chunk : strip_items;
                        got = true;
                        break;
                    }
                    strip_try++;
                }
                if (!got) { chunk_dry = true; continue; }
            }
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            const uint32_t avail = chunk_end - chunk_next, asked = (uint32_t)__popcll(need);
            const uint32_t first = chunk_next;
            chunk_next += asked < avail ? asked : avail;
            if (want && rank < avail) {
                const uint32_t mine = first + rank;
                const uint32_t th_log = c->tile_h_log, tw_log = 6u - th_log;
                const uint32_t tile = mine >> 6, l = mine & 63u;
                const uint32_t sl = tile % c->cam.s_count;
                const uint32_t q = tile / c->cam.s_count;
                const uint32_t tx = strip_lo + q % strip_w;
                const uint32_t k = q / strip_w;
                const uint32_t tiles_y = (c->cam.nrows + (1u << th_log) - 1u) >> th_log;
                const uint32_t ty = (k & 1u) ? tiles_y / 2 - (k + 1) / 2 : tiles_y / 2 + k / 2;
                const uint32_t x = (tx << tw_log) + (l & ((1u << tw_log) - 1u)), r = (ty << th_log) + (l >> tw_log);
                if (x >= c->cam.width || r >= c->cam.nrows) continue;   // padding of the last tile column / row
                item = (sl * c->cam.nrows + r) * c->cam.width + x;
                const uint32_t s = c->cam.s_first + sl;
                const uint32_t ci = c->cam.x0 + x;
                const uint32_t cj = c->cam.y0 + c->cam.row_first + r * c->cam.row_stride;
                n_paths++;
                // castRay prologue                      IntegratorDeltaTracking.cpp:13-52
                const f3 wd = primary_direction<EXACT>(c->cam, ci, cj, s);
                wdir = wd;
                const f3 weye = mk3(c->cam.pos[0], c->cam.pos[1], c->cam.pos[2]);
                const f3 ieye = mat_mult(c->inv_mat, mk3(weye.x - c->vec[0], weye.y - c->vec[1], weye.z - c->vec[2]));
                const f3 idir = mat_mult(c->inv_mat, wd);
                const float length = M::sqrt_(idir.x * idir.x + idir.y * idir.y + idir.z * idir.z);
                const float inv_length = M::rcp_(length);
                const f3 ndir = mk3(inv_length * idir.x, inv_length * idir.y, inv_length * idir.z);
                float c0 = length * 1e-5f, c1 = VR_FLT_MAX;
                int  seg = SEG_FOUND;
                bool seg_empty = false;
                const bool in_box = ray_clip<EXACT>(ieye, ndir, c0, c1, c->bmin, c->bmax1);
                if (MESH && in_box) {
                    medium = -1;
                    seg = mesh_clip(c->mesh, weye, wd, length, -1, medium, c0, c1, seg_empty, n_queries);
                }
                if (MESH && in_box && seg == SEG_BLACK) {
                    store_sample(p.work.samples, item, mk3(0, 0, 0));                         // an opaque surface
                } else if (!in_box || c->max_depth == 0 || (MESH && (seg == SEG_NONE || seg_empty))) {
                    store_sample(p.work.samples, item, mk3(c->background[0], c->background[1], c->background[2]));   // :37-39
                } else {
                    st = DD_WALK;
                    depth_remaining = (int)c->max_depth;
                    rng_pixel = (cj << 16) | ci; rng_ts = s; steps = 0;
                    seed_stream();
                    set_ray(ieye, ndir, c0, c1);
                }
            }
        }
        if (!__any(st == DD_WALK)) break;

        // =========================== step loop ===========================
        if (st == DD_WALK && steps > p.work.step_limit) { n_limit++; st = DD_BLACK; }
#pragma unroll 1
        for (int it = 0; it < p.round_max; ++it) {
            if (st == DD_WALK) {
                // the brick's far face (or the end of the ray)
                const float tface = fminf(fminf(tmx, tmy), tmz);
                const float te = fminf(tface, t1);
                float t_next = __builtin_huge_valf();
                uint32_t w_event = 0;
                if (mu > 0.0f) {
                    uint32_t w_path;
                    xoshiro128p_next2(xs0, xs1, xs2, xs3, w_path, w_event);
                    if (EXACT) {
                        t_next = t + M::div_(-M::log_(in_range(w_path, 0.00001f, 1.0f)), mu);          // getFloat(0.00001, 1), :63
                    } else {
                        const float u = fmaf((float)(w_path >> 8), (1.0f - 0.00001f) / 16777216.0f, 0.00001f);
                        t_next = fmaf(__builtin_amdgcn_logf(u), -0.6931471805599453f * __builtin_amdgcn_rcpf(mu), t);
                    }
                }
                if (!(t_next < te)) {
                    // no collision inside this brick: on to the next one
                    t = te;
                    if (tface >= t1) st = DD_ESCAPED;                                         // :77-79 the ray has left the grid
                    else {
                        if (tmx <= tmy && tmx <= tmz) { bi += dir.x > 0 ? 1 : -1; tmx += tdx; }
                        else if (tmy <= tmz)          { bj += dir.y > 0 ? 1 : -1; tmy += tdy; }
                        else                          { bk += dir.z > 0 ? 1 : -1; tmz += tdz; }
                        if ((uint32_t)bi >= (uint32_t)p.nc[0] || (uint32_t)bj >= (uint32_t)p.nc[1] || (uint32_t)bk >= (uint32_t)p.nc[2])
                            st = DD_ESCAPED;                                                  // (rounding: the last face and t1 coincide)
                        else mu = major_at(((uint32_t)bi * (uint32_t)p.nc[1] + (uint32_t)bj) * (uint32_t)p.nc[2] + (uint32_t)bk) * mu_scale;
                    }
                } else {
                    // tentative collision at t_next: real with probability mu_t(x) / mu
                    t = t_next;
                    steps++;
                    // :55, at the collision.  The voxel index is clamped into the DDA's cell: on a face rounding may put
                    // floor(ray(t)) into the neighbour, whose density the cell's majorant does not bound; and into the bbox
                    // (a cell may stick out of it; the ray's interval does not)
                    float sigma;
                    {
                        const f3      q = ray_at(eye, dir, t);
                        const int32_t span = (1 << cl) - 1;
                        int32_t vi = floor_to_int(q.x) - p.density.min[0], vj = floor_to_int(q.y) - p.density.min[1],
                                vk = floor_to_int(q.z) - p.density.min[2];
                        vi = max(bi << cl, min(vi, min((bi << cl) + span, p.density.dim[0] - 1)));
                        vj = max(bj << cl, min(vj, min((bj << cl) + span, p.density.dim[1] - 1)));
                        vk = max(bk << cl, min(vk, min((bk << cl) + span, p.density.dim[2] - 1)));
                        sigma = p.density.bricks[brick_offset(p.density, (uint32_t)vi, (uint32_t)vj, (uint32_t)vk)];
                    }
                    sigma *= p.light_mult;
                    const float thr = (float)(w_event >> 8) * (1.0f / 16777216.0f) * mu;
                    const float mu_a = sigma * p.sigma_a;
                    if (thr < mu_a) { absorbed_mu_a = mu_a; st = DD_ABSORBED; }              // :101-113
                    else if (thr < sigma * sig_t) st = DD_SCATTER;                            // :115
                }
            }
            const int n_walk = __popcll(__ballot(st == DD_WALK));
            w_trips++; w_trip_lanes += (unsigned)n_walk;
            const int n_event = __popcll(__ballot(st >= DD_SCATTER));
            if (n_event >= p.round_events || n_walk == 0) break;
        }
    }

    unsigned long long a = wave_sum(n_paths), b = wave_sum(n_lookups), c = wave_sum(n_scatter), e = wave_sum(n_limit);
    const unsigned long long nq = MESH ? wave_sum(n_queries) : 0ull;
    if (lane == 0) {
        if (MESH) atomicAdd(&p.work.counters->ray_queries, nq);
        atomicAdd(&p.work.counters->path_samples, a);
        atomicAdd(&p.work.counters->density_lookups, b);
        atomicAdd(&p.work.counters->voxel_loads, b);
        atomicAdd(&p.work.counters->scatter_events, c);
        if (e) atomicAdd(&p.work.counters->step_limit_hits, e);
        atomicAdd(&p.work.counters->wave_trips, w_trips);
        atomicAdd(&p.work.counters->wave_trip_lanes, w_trip_lanes);
        atomicAdd(&p.work.counters->event_phases, w_phases);
    }
}

// =============================================================================================
// mesh-bounded fixed-step marchers
// =============================================================================================

struct MarchParams {
    CameraDev cam;
    WorkDev   work;
    MeshDev   mesh;
    float     sigma_a, sigma_s, g;
    float     light[3], light_color[3], background[3];
    uint32_t  max_depth;
    uint32_t  tile_h_log;
};

// KIND: which handleIntersection marches the segment between entry and exit hit
enum { MARCH_NEE = 0, MARCH_IMPROVED = 1, MARCH_PERLIN = 2, MARCH_DENSITY = 3 };

// Waves per SIMD the register allocator must allow for the marchers (0 = whatever the kernel needs: 86-130 VGPRs, five to
// three waves).  With the wave-uniform BVH walk a query is a chain of scalar fetch -> slab test -> next node, and what
// hides the fetch is other waves: seven per SIMD (72 VGPRs, 60-170 bytes of scratch outside the walk) measured against
// the natural allocation on config 5 — sphere 90.6 -> 79.1 ms (Improved) and 85.6 -> 67.9 (NEE), CornellBox-Sphere
// 50.7 -> 47.6 and 47.3 -> 41.6, Perlin marcher 15.5 -> 12.7; six and eight are 2-9 % behind seven.  (Before that walk
// the kernel was bound by the L1 request path and forced occupancy bought nothing: DESIGN.md 6b.)
#ifndef VR_MARCH_WAVES
#define VR_MARCH_WAVES 7
#endif
template <int KIND, bool EXACT>
__global__ void __launch_bounds__(256)
#if VR_MARCH_WAVES
__attribute__((amdgpu_waves_per_eu(VR_MARCH_WAVES, VR_MARCH_WAVES)))
#endif
vr_march_kernel(const MarchParams p)
{
    typedef Math<EXACT> M;
    constexpr bool IMPROVED = KIND == MARCH_IMPROVED;
    unsigned long long n_lookups = 0;
    const uint32_t k0 = p.work.key0, k1 = p.work.key1;
    uint32_t n_queries = 0, n_light = 0, n_paths = 0, n_guard = 0;
    // a wave takes an 8x8-pixel tile (p.work.nitems counts whole tiles): its 64 rays visit nearly the same BVH nodes
    // and triangles, which a wave executes as the union over its lanes (+10-17 % against 64 pixels of a row)
    // (the tile is (64 >> tile_h_log) x (1 << tile_h_log) pixels of the row set: flatter when the rows are a rank's
    // share of an interleaved frame, so that it stays compact in the image)
    const uint32_t th_log = p.tile_h_log, tw_log = 6u - th_log;
    const uint32_t tiles_x = (p.cam.width + (1u << tw_log) - 1u) >> tw_log, tiles_y = (p.cam.nrows + (1u << th_log) - 1u) >> th_log;
    for (uint32_t mine = blockIdx.x * blockDim.x + threadIdx.x; mine < p.work.nitems; mine += gridDim.x * blockDim.x) {
        const uint32_t tile = mine >> 6, l = mine & 63u;
        const uint32_t tx = tile % tiles_x;
        const uint32_t q = tile / tiles_x;
        const uint32_t ty = q % tiles_y;
        const uint32_t sl = q / tiles_y;
        const uint32_t x = (tx << tw_log) + (l & ((1u << tw_log) - 1u)), r = (ty << th_log) + (l >> tw_log);
        if (x >= p.cam.width || r >= p.cam.nrows) continue;   // padding of the last tile column / row (no wave-level operation below)
        const uint32_t item = (sl * p.cam.nrows + r) * p.cam.width + x;
        const uint32_t s = p.cam.s_first + sl;
        const uint32_t ci = p.cam.x0 + x;
        const uint32_t cj = p.cam.y0 + p.cam.row_first + r * p.cam.row_stride;
        n_paths++;
        const uint32_t rng_pixel = (cj << 16) | ci;

        f3  origin = mk3(p.cam.pos[0], p.cam.pos[1], p.cam.pos[2]);
        f3  dir = primary_direction(p.cam, ci, cj, s);
        int object_id = -1, previous_id = -1;                                  // BaseIntegrator.cpp:63
        f3  result = mk3(0, 0, 0);
        const f3 bg = mk3(p.background[0], p.background[1], p.background[2]);
        uint32_t depth = 0;

        // castRay + handleIntersection, recursion unrolled  IntegratorHomogeneousRayMarcherNEE.cpp:10-118
        int guard = 0;
        for (; guard < 4096; ++guard) {
            if (depth >= p.max_depth) { result = mk3(0, 0, 0); break; }         // :12-14
            float tfar;
            n_queries++;
            const int prim = closest_hit_march(p.mesh, origin, dir, tfar);            // :25
            if (prim < 0) { result = bg; break; }                               // :44
            previous_id = object_id;                                            // :30
            object_id = (int)p.mesh.prim_to_shape[prim];                        // :33
            const f3 hit_point = add3(scale3(dir, tfar), origin);               // :34
            if (KIND == MARCH_DENSITY) {
                // IntegratorDensitySampling::handleIntersection (IntegratorDensitySampling.cpp:46-65): analytic
                // Beer-Lambert between the entry and exit hit of any object, tinted by its diffuse colour
                if (previous_id == object_id) {
                    const float transmission = M::exp_(-tfar * p.sigma_a);     // :55
                    f3 colour = mk3(0, 0, 0);
                    if (p.mesh.shape_diffuse) colour = mk3(p.mesh.shape_diffuse[3 * object_id], p.mesh.shape_diffuse[3 * object_id + 1],
                                                           p.mesh.shape_diffuse[3 * object_id + 2]);
                    result = add3(scale3(bg, transmission), scale3(colour, 1 - transmission));   // :57
                    break;
                }
                origin = add3(hit_point, scale3(dir, 0.001f));                  // :62
                continue;
            }
            const float dissolve = p.mesh.shape_dissolve[object_id];            // :49
            if (dissolve > 0) {                                                 // :52
                if (KIND == MARCH_PERLIN && previous_id == object_id) {
                    // IntegratorHeterogeneousPerlinNoise::handleIntersection (IntegratorHeterogeneousPerlinNoise.cpp:62-153):
                    // density from PerlinNoiseSampler::eval_density at every sample and along every light ray;
                    // the step is NOT rescaled to distance / ns here
                    float       transparency = 1;
                    f3          acc = mk3(0, 0, 0);
                    const float distance = tfar;                                // :64
                    const f3    light_dir = mk3(p.light[0], p.light[1], p.light[2]);
                    const f3    light_color = mk3(p.light_color[0], p.light_color[1], p.light_color[2]);
                    const float sigma_t = p.sigma_a + p.sigma_s;                // :74
                    const float step_size = 0.2f;                               // :78
                    const int   ns = (int)ceilf(distance / step_size);          // :79
                    const f3    nd = normalize3(dir);                           // :86
                    for (int n = 0; n < ns; n++) {                              // :89
                        uint32_t w0, w1, w2, w3;   // one Philox block per march step: jitter, roulette
                        philox4x32_10(rng_pixel, s, 0u, (uint32_t)n, k0, k1, w0, w1, w2, w3);
                        const float tt = (float)(step_size * ((double)n + (double)unit24(w0)));   // :91-99
                        const f3    sp = add3(origin, scale3(nd, tt));          // :101
                        const float density = perlin_eval_density(sp);          // :103
                        n_lookups++;
                        transparency *= M::exp_(-step_size * density * sigma_t);   // :106-109
                        const f3 lo = add3(sp, scale3(light_dir, 0.001f));      // :112-113
                        float    dl = 0;
                        bool     lit = false;
                        if (density > 0) { n_queries++; n_light++; lit = closest_hit_march(p.mesh, lo, light_dir, dl) >= 0; }   // :116-117
                        if (lit) {
                            float       tau = 0;
                            const float num_steps_light = ceilf(dl / step_size);   // :120
                            for (uint32_t nl = 0; (float)nl < num_steps_light; ++nl) {   // :122
                                const float t_light = (float)(step_size * ((double)nl + 0.5));   // :123
                                tau += perlin_eval_density(add3(sp, scale3(light_dir, t_light)));   // :124-125
                                n_lookups++;
                            }
                            const float cos_theta = dot3(nd, light_dir);        // :128
                            const float la = M::exp_(-tau * step_size * sigma_t);   // :129
                            f3 c = scale3(light_color, la);                     // :130-137
                            c = scale3(c, density);
                            c = scale3(c, p.sigma_s);
                            c = scale3(c, hg_phase(p.g, cos_theta));
                            c = scale3(c, transparency);
                            c = scale3(c, step_size);
                            acc = add3(acc, c);
                            if ((double)transparency < 1e-3) {                  // :140-148
                                if (in_range(w1, 0, 1) > 1.f / 2) n = ns;
                                else transparency *= 2;
                            }
                        }
                    }
                    result = add3(scale3(bg, transparency), acc);               // :153
                    break;
                }
                if (KIND != MARCH_PERLIN && previous_id == object_id) {         // :55 exit hit: march the segment
                    float       transparency = 1;
                    f3          acc = mk3(0, 0, 0);
                    const float distance = tfar;                                // :59
                    const f3    light_dir = mk3(p.light[0], p.light[1], p.light[2]);
                    const f3    light_color = mk3(p.light_color[0], p.light_color[1], p.light_color[2]);
                    const float density = 1.0f;                                 // Improved :69
                    float       step_size = 0.2f;                               // :66-68
                    const int   ns = (int)ceilf(distance / step_size);
                    step_size = distance / ns;
                    const f3 nd = normalize3(dir);                              // :70
                    for (int n = 0; n < ns; n++) {                              // :73
                        float tt;
                        uint32_t w0 = 0, w1 = 0, w2, w3;   // one Philox block per march step: jitter, roulette
                        if (IMPROVED) {
                            philox4x32_10(rng_pixel, s, 0u, (uint32_t)n, k0, k1, w0, w1, w2, w3);
                            const double off = (double)unit24(w0);              // Improved :82-84
                            tt = (float)(step_size * ((double)n + off));        // Improved :90
                        } else {
                            tt = step_size * ((float)n + 0.5f);                 // :75
                        }
                        const f3    sp = add3(origin, scale3(nd, tt));          // :76
                        const float att = IMPROVED ? M::exp_(-step_size * density * (p.sigma_a + p.sigma_s))
                                                   : M::exp_(-step_size * p.sigma_a); // :79
                        transparency *= att;                                    // :82
                        const f3 lo = add3(sp, scale3(light_dir, 0.001f));      // :84-85
                        float    dl;
                        n_queries++; n_light++;
                        if (closest_hit_march(p.mesh, lo, light_dir, dl) >= 0) {      // :88
                            if (IMPROVED) {
                                const float cos_theta = dot3(nd, light_dir);    // Improved :107
                                const float la = M::exp_(-dl * density * (p.sigma_a + p.sigma_s));
                                f3 c = scale3(light_color, la);                 // Improved :110-117
                                c = scale3(c, density);
                                c = scale3(c, p.sigma_s);
                                c = scale3(c, hg_phase(p.g, cos_theta));
                                c = scale3(c, transparency);
                                c = scale3(c, step_size);
                                acc = add3(acc, c);
                            } else {
                                const float la = M::exp_(-dl * p.sigma_a);      // :91
                                f3 c = scale3(light_color, transparency);       // :93-97
                                c = scale3(c, la);
                                c = scale3(c, step_size);
                                acc = add3(acc, c);
                            }
                        }
                        if (IMPROVED) {                                         // Improved :121-129
                            if ((double)transparency < 1e-3) {
                                if (in_range(w1, 0, 1) > 1.f / 2) n = ns;
                                else transparency *= 2;
                            }
                        }
                    }
                    result = add3(scale3(bg, transparency), acc);               // :102
                    break;
                }
                origin = add3(hit_point, scale3(dir, 0.001f));                  // :105
                continue;
            }
            if (previous_id == object_id) {                                     // :112-117
                origin = add3(origin, scale3(dir, 0.001f));
                depth += 1;
                continue;
            }
            result = mk3(0, 0, 0);   // the reference falls off a non-void function here (:118): black
            break;
        }
        if (guard == 4096) n_guard++;   // where the reference would still recurse: the sample stays black and is reported
        store_sample(p.work.samples, item, result);
    }
    unsigned long long a = wave_sum(n_paths), b = wave_sum(n_queries), c = wave_sum(n_light), e = wave_sum(n_lookups);
    const unsigned long long gh = wave_sum(n_guard);
    if ((threadIdx.x & 63) == 0) {
        if (gh) atomicAdd(&p.work.counters->step_limit_hits, gh);
        atomicAdd(&p.work.counters->path_samples, a);
        atomicAdd(&p.work.counters->ray_queries, b);
        atomicAdd(&p.work.counters->shadow_rays, c);
        if (KIND == MARCH_PERLIN) atomicAdd(&p.work.counters->density_lookups, e);
    }
}

// =============================================================================================
// fixed-step marchers through the voxel grid
// =============================================================================================
// IntegratorNanoVDBSimple::castRay (IntegratorNanoVDBSimple.cpp:10-95) and
// IntegratorNanoVDBEmission::castRay (IntegratorNanoVDBEmission.cpp:10-110): half-voxel steps along the
// clipped index-space ray, at every non-empty step a 10-voxel-step march toward the light.  One path
// sample per lane, grid-stride; deterministic but for the roulette at transmittance < 1e-3.
struct GridMarchParams {
    CameraDev cam;
    WorkDev   work;
    GridDev   density, temperature;
    float     inv_mat[9], vec[3], mat[9];
    float     bmin[3], bmax1[3];
    float     g;
    float     light[3], light_color[3], emission[3], background[3];
};

#ifndef VR_GRIDMARCH_WAVES
#define VR_GRIDMARCH_WAVES 0   // waves per SIMD the register allocator must allow (0: what the kernel needs, 73-78 VGPRs = six)
#endif
template <bool EMISSION, bool EXACT>
__global__ void __launch_bounds__(256)
#if VR_GRIDMARCH_WAVES
__attribute__((amdgpu_waves_per_eu(VR_GRIDMARCH_WAVES, VR_GRIDMARCH_WAVES)))
#endif
vr_gridmarch_kernel(const GridMarchParams p)
{
    typedef Math<EXACT> M;
    const uint32_t k0 = p.work.key0, k1 = p.work.key1;
    unsigned long long n_lookups = 0;
    uint32_t n_paths = 0, n_light = 0;
    const f3 light = mk3(p.light[0], p.light[1], p.light[2]);
    const f3 light_color = mk3(p.light_color[0], p.light_color[1], p.light_color[2]);
    const f3 bg = mk3(p.background[0], p.background[1], p.background[2]);
    // the light ray's index-space direction does not depend on the sample        Ray.h:257-267
    const f3    l_idir = mat_mult(p.inv_mat, light);
    const float l_length = sqrtf(l_idir.x * l_idir.x + l_idir.y * l_idir.y + l_idir.z * l_idir.z);
    const float l_inv_length = 1.0f / l_length;
    const f3    l_dir = mk3(l_inv_length * l_idir.x, l_inv_length * l_idir.y, l_inv_length * l_idir.z);

    // Work: 8x8-pixel tiles, one per wave at a time, handed out by the launch's counter.  The lanes of a wave
    // then look up voxels a few apart in two directions instead of 26 apart in one (a 64-pixel row segment of
    // the 1080p frame spans 26 voxels of the 512^3 cloud and every lane pulled its own 64-byte line: 500 GB of
    // fabric traffic for 50 GB of voxels), and a wave that draws cheap pixels simply draws again.
    const int      lane = threadIdx.x & 63;
    const uint32_t tiles_x = (p.cam.width + 7u) / 8u, tiles_y = (p.cam.nrows + 7u) / 8u;
    // Tile order.  The light marches of vertically adjacent tiles run through the same voxel columns, and so
    // do the sub-samples of one tile; tiles drawn one after the other are rendered at the same time.  So the
    // frame is cut into 8 vertical strips, one per XCD (workgroups go round-robin over the XCDs, each with an
    // L2 of its own), and a strip goes out column by column, within a column sample by sample, rows innermost.
    // A wave whose strip is used up helps with the next one.
    const uint32_t column_tiles = tiles_y * p.cam.s_count;
#ifndef VR_GM_STRIPS
#define VR_GM_STRIPS 8
#endif
    uint32_t strip_try = 0;
    const uint32_t xcd = VR_GM_STRIPS > 1 ? (blockIdx.x & 7u) : 0u;
    for (;;) {
        uint32_t tile = 0, c_lo = 0, c_hi = 0;
        bool     got = false;
        while (strip_try < VR_GM_STRIPS) {
            const uint32_t strip = (xcd + strip_try) % VR_GM_STRIPS;
            c_lo = tiles_x * strip / VR_GM_STRIPS; c_hi = tiles_x * (strip + 1u) / VR_GM_STRIPS;
            if (lane == 0) tile = atomicAdd(p.work.counter + strip * 8u, 1u);
            tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
            if (tile < (c_hi - c_lo) * column_tiles) { got = true; break; }
            strip_try++;
        }
        if (!got) break;
        const uint32_t tx = c_lo + tile / column_tiles, rem = tile % column_tiles;
        const uint32_t sl = rem / tiles_y, ty = rem - sl * tiles_y;
        const uint32_t x = tx * 8u + ((uint32_t)lane & 7u), r = ty * 8u + ((uint32_t)lane >> 3);
        if (x >= p.cam.width || r >= p.cam.nrows) continue;   // (no wave-level operation in the body below)
        const uint32_t mine = (sl * p.cam.nrows + r) * p.cam.width + x;
        const uint32_t s = p.cam.s_first + sl;
        const uint32_t ci = p.cam.x0 + x;
        const uint32_t cj = p.cam.y0 + p.cam.row_first + r * p.cam.row_stride;
        n_paths++;
        const uint32_t rng_pixel = (cj << 16) | ci;

        const f3 wdir = primary_direction(p.cam, ci, cj, s);
        const f3 ray_direction = normalize3(wdir);                                  // Simple :12
        const f3 weye = mk3(p.cam.pos[0], p.cam.pos[1], p.cam.pos[2]);
        const f3 ieye = mat_mult(p.inv_mat, mk3(weye.x - p.vec[0], weye.y - p.vec[1], weye.z - p.vec[2]));   // :15-20
        const f3 idir = mat_mult(p.inv_mat, wdir);
        const float length = sqrtf(idir.x * idir.x + idir.y * idir.y + idir.z * idir.z);
        const float inv_length = 1.0f / length;
        const f3 ndir = mk3(inv_length * idir.x, inv_length * idir.y, inv_length * idir.z);
        float c0 = length * 1e-5f, c1 = VR_FLT_MAX;
        f3 result = bg;
        if (ray_clip(ieye, ndir, c0, c1, p.bmin, p.bmax1)) {                        // :22-24
            const float density = EMISSION ? 64.0f : 0.5f;                          // Simple :25, Emission :35
            const float light_ray_density = density * 0.5f;                         // :26
            const float step_size = 0.5f;                                           // :28
            float transmittance = 1.0f;                                             // :30
            f3 acc = mk3(0, 0, 0);                                                  // :32
            uint32_t n = 0;
            for (float t = c0; t < c1; t += step_size, ++n) {                       // :34
                const f3 ip = ray_at(ieye, ndir, t);
                const float sigma = grid_value_nocache(p.density, ip) * density;    // :36
                n_lookups++;
                float emission_value = 0.0f;
                if (EMISSION) {                                                     // Emission :47-48
                    emission_value = grid_value_nocache(p.temperature, ip) * 4.0f;
                    n_lookups++;
                }
                transmittance *= M::exp_(-step_size * sigma);                       // :38-40
                if (sigma > 0 || (EMISSION && emission_value > 0)) {                // Simple :47, Emission :61
                    // castLightRay (:97-128): from the sample's WORLD position (:42-45) toward the light
                    const f3 world = mk3(fmaf(ip.x, p.mat[0], fmaf(ip.y, p.mat[1], fmaf(ip.z, p.mat[2], p.vec[0]))),
                                         fmaf(ip.x, p.mat[3], fmaf(ip.y, p.mat[4], fmaf(ip.z, p.mat[5], p.vec[1]))),
                                         fmaf(ip.x, p.mat[6], fmaf(ip.y, p.mat[7], fmaf(ip.z, p.mat[8], p.vec[2]))));
                    const f3 leye = mat_mult(p.inv_mat, mk3(world.x - p.vec[0], world.y - p.vec[1], world.z - p.vec[2]));
                    float l0 = l_length * 1e-5f, l1 = VR_FLT_MAX;
                    n_light++;
                    if (ray_clip(leye, l_dir, l0, l1, p.bmin, p.bmax1)) {           // :49
                        float tau = 0;
                        const float light_step_size = step_size * 20.0f;            // :52
                        const float num_steps_light = ceilf(l1 / light_step_size);  // :53 (t1, not t1 - t0)
                        for (uint32_t nl = 0; (float)nl < num_steps_light; ++nl) {  // :55
                            // light_step_size * (nl + 0.5) in double, then float (:56): 10 nl + 5, an integer below 2^24,
                            // so the integer form is the same number without four fp64 instructions per step
                            const float t_light = nl < (1u << 20) ? (float)(10u * nl + 5u)
                                                                  : (float)((double)light_step_size * ((double)nl + 0.5));
                            tau += grid_value_nocache(p.density, ray_at(leye, l_dir, l0 + t_light)) * light_ray_density;   // :57
                            n_lookups++;
                        }
                        const float cos_theta = dot3(ray_direction, light);         // :60
                        const float light_attenuation = M::exp_(-tau * light_step_size * sigma);   // :61
                        f3 c = scale3(light_color, light_attenuation);              // :63-69
                        c = scale3(c, sigma);
                        c = scale3(c, hg_phase(p.g, cos_theta));
                        c = scale3(c, transmittance);
                        c = scale3(c, step_size);
                        acc = add3(acc, c);
                    } else {                                                        // :73-78
                        f3 c = scale3(light_color, sigma);
                        c = scale3(c, transmittance);
                        c = scale3(c, step_size);
                        acc = add3(acc, c);
                    }
                    if (EMISSION) {                                                 // Emission :48, :90
                        const f3 emission = scale3(mk3(p.emission[0], p.emission[1], p.emission[2]), emission_value);
                        acc = add3(acc, scale3(scale3(emission, transmittance), step_size));
                    }
                    if ((double)transmittance < 1e-3) {                             // :81-89
                        uint32_t w0, w1, w2, w3;
                        philox4x32_10(rng_pixel, s, 0u, n, k0, k1, w0, w1, w2, w3);
                        if (in_range(w1, 0, 1) > 1.f / 2) t = c1;
                        else transmittance *= 2;
                    }
                }
            }
            result = add3(acc, scale3(bg, transmittance));                          // :94
        }
        store_sample(p.work.samples, mine, result);
    }
    unsigned long long a = wave_sum(n_paths), b = wave_sum(n_lookups), c = wave_sum(n_light);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&p.work.counters->path_samples, a);
        atomicAdd(&p.work.counters->density_lookups, b);
        atomicAdd(&p.work.counters->voxel_loads, b);      // the grid marchers load a voxel per lookup
        atomicAdd(&p.work.counters->shadow_rays, c);
    }
}

// =============================================================================================
// resolve: color += castRay(...) for s = 0..spp-1, then color / spp            BaseIntegrator.cpp:71,92,108
// =============================================================================================
// accum[pixel] carries the running sum across passes; the last pass divides.
__global__ void __launch_bounds__(256) vr_resolve_kernel(const float* __restrict__ samples, float* __restrict__ accum,
                                                         uint32_t npix, uint32_t s_count, int first_pass, int last_pass,
                                                         float spp)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;   // one float channel per thread: coalesced
    const uint32_t n = npix * 3u;
    if (i >= n) return;
    float c = first_pass ? 0.0f : accum[i];
    for (uint32_t s = 0; s < s_count; ++s) c += samples[(size_t)s * n + i];
    accum[i] = last_pass ? c / spp : c;
}

} // namespace vr
