/*
 * vr_types.h — plain-old-data types shared by the C-ABI (vr_backend.h), the CPU oracle
 * (oracle/vr_oracle.h) and the host-side mirrors.  No algorithm lives here: only the
 * layout of what crosses the boundary.
 *
 * Every field cites the member of the reference it replaces
 * (paths relative to /root/reference/VolumeRenderer/RayTracer/).
 */
#ifndef VR_TYPES_H
#define VR_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Integrator selection.  The reference selects by string in
 * integrators/IntegratorFactory.cpp:12-24; the names below are the accepted strings. */
typedef enum vr_integrator_id {
    VR_INTEGRATOR_HOMOGENEOUS_RAY_MARCHER_NEE      = 0, /* "homogeneousRayMarcherNEE"      */
    VR_INTEGRATOR_HOMOGENEOUS_RAY_MARCHER_IMPROVED = 1, /* "homogeneousRayMarcherImproved" */
    VR_INTEGRATOR_RATIO_TRACKING                   = 2, /* "ratioTracking"                 */
    VR_INTEGRATOR_DELTA_TRACKING                   = 3, /* "deltaTracking"                 */
    VR_INTEGRATOR_NANOVDB_SIMPLE                   = 4, /* "nanoVDBSimple"                 */
    VR_INTEGRATOR_NANOVDB_EMISSION                 = 5, /* "nanoVDBEmission"               */
    VR_INTEGRATOR_HETEROGENEOUS_PERLIN_NOISE       = 6, /* "heterogeneousPerlinNoise"      */
    VR_INTEGRATOR_DENSITY_SAMPLING                 = 7, /* "densitySampling"               */
    /* No reference counterpart (its trackers clip to the grid bbox only, IntegratorDeltaTracking.cpp:37-39):
     * the trackers with their walk cut to the mesh's entry/exit interval — SURVEY 8(d) config 5. */
    VR_INTEGRATOR_MESH_RATIO_TRACKING              = 8, /* "meshRatioTracking"             */
    VR_INTEGRATOR_MESH_DELTA_TRACKING              = 9, /* "meshDeltaTracking"             */
    VR_INTEGRATOR_COUNT                            = 10
} vr_integrator_id;

/* POD mirror of `struct Options` (Utils/Types.h:85-122) without the strings/vectors.
 * Defaults (vr_options_defaults) follow Utils/XMLManager.cpp:39-107.  Vec3f members are
 * float[3] in (x, y, z) order.  `cameraRotation` keeps the reference's meaning:
 * [0]=pitch attr, [1]=yaw attr, [2]=roll attr (XMLManager.cpp:75-77). */
typedef struct vr_options_pod {
    float    sigma_s;                    /* Types.h:89  */
    float    sigma_a;                    /* Types.h:90  */
    float    sigma_n;                    /* Types.h:91 (parsed, unused by the path) */
    uint32_t rayPerPixelCount;           /* Types.h:92  */
    uint32_t width;                      /* Types.h:93  */
    uint32_t widthStartOffset;           /* Types.h:94  */
    uint32_t widthReference;             /* Types.h:95  */
    uint32_t height;                     /* Types.h:96  */
    uint32_t heightStartOffset;          /* Types.h:97  */
    uint32_t heightReference;            /* Types.h:98  */
    float    fov;                        /* Types.h:99  */
    uint32_t maxDepth;                   /* Types.h:101 (uint8_t there; must be <= 255) */
    float    backgroundColor[3];         /* Types.h:102 (already multiplied, XMLManager.cpp:67) */
    uint32_t multiThreadedChunkSize;     /* Types.h:106; only seeds the reference-like CPU RNG */
    float    cameraPosition[3];          /* Types.h:108 */
    float    cameraRotation[3];          /* Types.h:109 */
    float    lightPosition[3];           /* Types.h:111 (used as a direction) */
    float    lightColor[3];              /* Types.h:112 */
    float    emissionColor[3];           /* Types.h:113 */
    float    mediumColor[3];             /* Types.h:114 */
    float    heyneyGreensteinG;          /* Types.h:115 */
    uint32_t useImportanceSampling;      /* Types.h:116 (bool) */
    float    stepSizeMin;                /* Types.h:117 */
    float    stepSizeMax;                /* Types.h:118 */
    float    stepSizeMultiplier;         /* Types.h:119 */
    float    lightRayDensityMultiplier;  /* Types.h:120 */
    float    shadowRayDensityMultiplier; /* Types.h:121 */
} vr_options_pod;

/* One scalar float grid, the dense stand-in for `nanovdb::FloatGrid*` +
 * `nanovdb::Extrema<float>` + `nanovdb::CoordBBox` in SceneInfo (Utils/Types.h:141-145).
 *
 * voxels : (bbox_max - bbox_min + 1) values per axis, k (z) fastest, then j (y), then i (x):
 *          value(i,j,k) = voxels[((i-min0)*ny + (j-min1))*nz + (k-min2)].
 *          Host pointer; copied at set time.
 * bbox   : index-space, inclusive (tree().bbox(), SceneLoader.cpp:227-229).  Ray clipping
 *          pads max by one voxel (nanovdb/util/Ray.h:401).
 * inv_mat, vec : world->index map, index = inv_mat * (world - vec)
 *          (nanovdb/NanoVDB.h:2262-2270), row-major 3x3.
 * mat    : index->world map, world = mat * index + vec (Map::applyMapF, NanoVDB.h:2249): the
 *          Map's own mMatF, not a numerical inverse.  Only the fixed-step grid marchers
 *          (nanoVDBSimple / nanoVDBEmission) read it.
 * background : value returned outside the bbox (NanoVDB.h:3350).
 * max_value  : what nanovdb::getExtrema(...).max() returned (SceneLoader.cpp:219,223);
 *          pass a negative number to have it computed from the voxels. */
typedef struct vr_grid_desc {
    const float* voxels;
    int32_t      bbox_min[3];
    int32_t      bbox_max[3];
    float        inv_mat[9];
    float        vec[3];
    float        mat[9];
    float        background;
    float        max_value;
} vr_grid_desc;

/* Triangle scene, the stand-in for the Embree scene + `primitives` + materials of
 * SceneInfo (Utils/Types.h:130-133, SceneLoader.cpp:69-175).
 * prim_to_shape[t]   : SceneInfo::primitives[primID]  (SceneLoader.cpp:146)
 * shape_dissolve[s]  : materials[shapes[s].mesh.material_ids[0]].dissolve
 *                      (IntegratorHomogeneousRayMarcherNEE.cpp:49-52). */
typedef struct vr_mesh_desc {
    const float*    verts_xyz;      /* nverts * 3 */
    uint32_t        nverts;
    const uint32_t* tri_idx;        /* ntris * 3 */
    uint32_t        ntris;
    const uint32_t* prim_to_shape;  /* ntris */
    const float*    shape_dissolve; /* nshapes */
    const float*    shape_diffuse;  /* nshapes * 3: materials[...].diffuse (IntegratorDensitySampling.cpp:54);
                                       NULL = black; only densitySampling reads it */
    uint32_t        nshapes;
} vr_mesh_desc;

/* Work counters a render call can report (one lookup = one accessor getValue in the
 * reference: one per tracking step, one per shadow-march step, one per temperature read). */
typedef struct vr_render_stats {
    uint64_t path_samples;     /* castRay calls */
    uint64_t density_lookups;  /* getValue calls on the density (and temperature) grid */
    uint64_t scatter_events;
    uint64_t shadow_rays;      /* directLightningRayMarch calls / marcher light rays */
    uint64_t ray_queries;      /* closest-hit queries (rtcIntersect1 in the reference) */
    uint64_t step_limit_hits;  /* paths cut by the safety cap (0 in any valid run) */
    /* wave-level divergence counters of the GPU tracking kernel (0 from the CPU oracle) */
    uint64_t wave_trips;       /* trips of a wave through the step loop */
    uint64_t wave_trip_lanes;  /* lanes that were walking, summed over those trips (<= 64 per trip) */
    uint64_t event_phases;     /* event phases run by a wave (scatter / shadow end / refill) */
    /* density_lookups that issued a load on the GPU; the others found the voxel of the previous step
     * in the lane's register (sub-voxel steps) or were taken arithmetically as an empty-space run.
     * 4 B x voxel_loads is what the memory system is asked for.  0 from the CPU oracle. */
    uint64_t voxel_loads;
} vr_render_stats;

/* Error codes returned by every vr_* entry point (0 = success).  The reference has no
 * error convention (exit()/throw); these replace it at the boundary. */
enum {
    VR_OK                 = 0,
    VR_ERR_INVALID_ARG    = -1,
    VR_ERR_NO_DEVICE      = -2,
    VR_ERR_HIP            = -3,
    VR_ERR_NO_DENSITY     = -4, /* tracking integrator without a density grid (SceneLoader.cpp:212-213 throws) */
    VR_ERR_NO_MESH        = -5,
    VR_ERR_UNSUPPORTED    = -6,
    VR_ERR_STEP_LIMIT     = -7, /* a path hit the per-path step cap: stepSize too small for float t */
    VR_ERR_OUT_OF_MEMORY  = -8
};

#ifdef __cplusplus
}
#endif
#endif /* VR_TYPES_H */
