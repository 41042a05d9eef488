"""ctypes mirror of include/vr_types.h (layout only, no algorithm).

Field order and types must match the header; tests/test_abi.py checks sizeof against the
values the C library reports.
"""
import ctypes as C

import numpy as np

INTEGRATORS = {
    # names accepted by the reference's IntegratorFactory (integrators/IntegratorFactory.cpp:12-24)
    "homogeneousRayMarcherNEE": 0,
    "homogeneousRayMarcherImproved": 1,
    "ratioTracking": 2,
    "deltaTracking": 3,
    "nanoVDBSimple": 4,
    "nanoVDBEmission": 5,
    "heterogeneousPerlinNoise": 6,
    "densitySampling": 7,
    # no reference counterpart: the trackers cut to the mesh's entry/exit interval
    "meshRatioTracking": 8,
    "meshDeltaTracking": 9,
}

VR_OK = 0
VR_ERR_INVALID_ARG = -1
VR_ERR_NO_DEVICE = -2
VR_ERR_HIP = -3
VR_ERR_NO_DENSITY = -4
VR_ERR_NO_MESH = -5
VR_ERR_UNSUPPORTED = -6
VR_ERR_STEP_LIMIT = -7
VR_ERR_OUT_OF_MEMORY = -8

F3 = C.c_float * 3


class OptionsPod(C.Structure):
    """vr_options_pod — POD mirror of the reference's Options (Utils/Types.h:85-122)."""

    _fields_ = [
        ("sigma_s", C.c_float),
        ("sigma_a", C.c_float),
        ("sigma_n", C.c_float),
        ("rayPerPixelCount", C.c_uint32),
        ("width", C.c_uint32),
        ("widthStartOffset", C.c_uint32),
        ("widthReference", C.c_uint32),
        ("height", C.c_uint32),
        ("heightStartOffset", C.c_uint32),
        ("heightReference", C.c_uint32),
        ("fov", C.c_float),
        ("maxDepth", C.c_uint32),
        ("backgroundColor", F3),
        ("multiThreadedChunkSize", C.c_uint32),
        ("cameraPosition", F3),
        ("cameraRotation", F3),
        ("lightPosition", F3),
        ("lightColor", F3),
        ("emissionColor", F3),
        ("mediumColor", F3),
        ("heyneyGreensteinG", C.c_float),
        ("useImportanceSampling", C.c_uint32),
        ("stepSizeMin", C.c_float),
        ("stepSizeMax", C.c_float),
        ("stepSizeMultiplier", C.c_float),
        ("lightRayDensityMultiplier", C.c_float),
        ("shadowRayDensityMultiplier", C.c_float),
    ]


class GridDesc(C.Structure):
    """vr_grid_desc"""

    _fields_ = [
        ("voxels", C.POINTER(C.c_float)),
        ("bbox_min", C.c_int32 * 3),
        ("bbox_max", C.c_int32 * 3),
        ("inv_mat", C.c_float * 9),
        ("vec", F3),
        ("mat", C.c_float * 9),
        ("background", C.c_float),
        ("max_value", C.c_float),
    ]


class MeshDesc(C.Structure):
    """vr_mesh_desc"""

    _fields_ = [
        ("verts_xyz", C.POINTER(C.c_float)),
        ("nverts", C.c_uint32),
        ("tri_idx", C.POINTER(C.c_uint32)),
        ("ntris", C.c_uint32),
        ("prim_to_shape", C.POINTER(C.c_uint32)),
        ("shape_dissolve", C.POINTER(C.c_float)),
        ("shape_diffuse", C.POINTER(C.c_float)),
        ("nshapes", C.c_uint32),
    ]


class RenderStats(C.Structure):
    """vr_render_stats"""

    _fields_ = [
        ("path_samples", C.c_uint64),
        ("density_lookups", C.c_uint64),
        ("scatter_events", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("ray_queries", C.c_uint64),
        ("step_limit_hits", C.c_uint64),
        ("wave_trips", C.c_uint64),
        ("wave_trip_lanes", C.c_uint64),
        ("event_phases", C.c_uint64),
        ("voxel_loads", C.c_uint64),
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def default_options(**kw):
    """Options with the reference's XML defaults (Utils/XMLManager.cpp:39-107)."""
    o = OptionsPod()
    o.sigma_s = 0.5
    o.sigma_a = 0.5
    o.rayPerPixelCount = 1
    o.maxDepth = 4
    o.width = 400
    o.widthReference = 400
    o.height = 400
    o.heightReference = 400
    o.multiThreadedChunkSize = 20
    o.fov = 90.0
    o.stepSizeMin = 0.01
    o.stepSizeMax = 1.0
    o.stepSizeMultiplier = 1.0
    o.lightRayDensityMultiplier = 1.0
    o.shadowRayDensityMultiplier = 1.0
    set_options(o, **kw)
    return o


def set_options(o, **kw):
    for k, v in kw.items():
        cur = getattr(o, k)  # raises AttributeError on a typo
        if isinstance(cur, C.Array):
            vals = list(v)
            if len(vals) != len(cur):
                raise ValueError(f"{k}: expected {len(cur)} values")
            for i, x in enumerate(vals):
                cur[i] = x
        else:
            setattr(o, k, v)
    return o


def copy_options(o, **kw):
    c = OptionsPod()
    C.memmove(C.byref(c), C.byref(o), C.sizeof(OptionsPod))
    return set_options(c, **kw)


class Grid:
    """Owns the numpy voxel array behind a GridDesc (keeps it alive)."""

    def __init__(self, voxels, bbox_min=None, voxel_size=1.0, origin=(0.0, 0.0, 0.0), background=0.0,
                 max_value=-1.0, inv_mat=None, vec=None, mat=None):
        self.voxels = np.ascontiguousarray(voxels, dtype=np.float32)
        assert self.voxels.ndim == 3
        nx, ny, nz = self.voxels.shape
        if bbox_min is None:
            bbox_min = (-(nx // 2), -(ny // 2), -(nz // 2))
        self.bbox_min = tuple(int(v) for v in bbox_min)
        self.bbox_max = (self.bbox_min[0] + nx - 1, self.bbox_min[1] + ny - 1, self.bbox_min[2] + nz - 1)
        d = GridDesc()
        d.voxels = self.voxels.ctypes.data_as(C.POINTER(C.c_float))
        for i in range(3):
            d.bbox_min[i] = self.bbox_min[i]
            d.bbox_max[i] = self.bbox_max[i]
        if mat is None:
            if inv_mat is None:
                vsz = np.float32(voxel_size)
                mat = [vsz, 0, 0, 0, vsz, 0, 0, 0, vsz]
            else:   # callers that have the Map's own mMatF pass it; this is only a numerical stand-in
                mat = np.linalg.inv(np.asarray(inv_mat, dtype=np.float64).reshape(3, 3)).astype(np.float32).ravel()
        if inv_mat is None:
            inv = np.float32(1.0 / float(voxel_size))
            inv_mat = [inv, 0, 0, 0, inv, 0, 0, 0, inv]
        if vec is None:
            vec = origin
        for i in range(9):
            d.inv_mat[i] = inv_mat[i]
            d.mat[i] = mat[i]
        for i in range(3):
            d.vec[i] = vec[i]
        d.background = background
        d.max_value = max_value
        self.desc = d


class Mesh:
    """Owns the arrays behind a MeshDesc."""

    def __init__(self, verts, tris, prim_to_shape=None, shape_dissolve=None, shape_diffuse=None):
        self.verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        self.tris = np.ascontiguousarray(tris, dtype=np.uint32).reshape(-1, 3)
        nt = self.tris.shape[0]
        if prim_to_shape is None:
            prim_to_shape = np.zeros(nt, dtype=np.uint32)
        self.prim_to_shape = np.ascontiguousarray(prim_to_shape, dtype=np.uint32)
        if shape_dissolve is None:
            shape_dissolve = np.ones(int(self.prim_to_shape.max()) + 1 if nt else 1, dtype=np.float32)
        self.shape_dissolve = np.ascontiguousarray(shape_dissolve, dtype=np.float32)
        d = MeshDesc()
        d.verts_xyz = self.verts.ctypes.data_as(C.POINTER(C.c_float))
        d.nverts = self.verts.shape[0]
        d.tri_idx = self.tris.ctypes.data_as(C.POINTER(C.c_uint32))
        d.ntris = nt
        d.prim_to_shape = self.prim_to_shape.ctypes.data_as(C.POINTER(C.c_uint32))
        d.shape_dissolve = self.shape_dissolve.ctypes.data_as(C.POINTER(C.c_float))
        d.nshapes = self.shape_dissolve.shape[0]
        if shape_diffuse is not None:
            self.shape_diffuse = np.ascontiguousarray(shape_diffuse, dtype=np.float32).reshape(-1, 3)
            assert self.shape_diffuse.shape[0] == d.nshapes
            d.shape_diffuse = self.shape_diffuse.ctypes.data_as(C.POINTER(C.c_float))
        self.desc = d
