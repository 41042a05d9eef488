"""ctypes binding of the C-ABI (include/vr_backend.h -> csrc/libvr_backend.so).

This is host plumbing only: every compute call goes to the HIP library.  There is no CPU
fallback — a missing library or GPU raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .abi import GridDesc, MeshDesc, OptionsPod, RenderStats

MODE_EXACT = 0
MODE_FAST = 1
MODE_EXACT_SKIP = 2
MODE_FAST_LOCAL = 3
MODE_EXACT_LOCAL = 4

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvr_backend.so")

# every symbol include/vr_backend.h declares
EXPORTS = [
    "vr_backend_create", "vr_backend_destroy", "vr_backend_last_error", "vr_backend_set_density",
    "vr_backend_generate_cloud", "vr_backend_read_density", "vr_backend_density_max", "vr_backend_set_mesh",
    "vr_backend_set_mode", "vr_backend_set_seed", "vr_backend_set_sample_buffer_bytes", "vr_backend_render_rows",
    "vr_backend_render", "vr_backend_render_row_set_device", "vr_backend_synchronize", "vr_backend_last_stats",
    "vr_backend_tonemap_device", "vr_backend_max_channel_device", "vr_rgb_to_u8", "vr_abi_sizes",
    "vr_backend_render_row_set_device_noalloc", "vr_backend_set_density_map",
    "vr_group_create", "vr_group_destroy", "vr_group_size", "vr_group_backend", "vr_group_last_error", "vr_group_set_density",
    "vr_group_generate_cloud", "vr_group_set_density_map", "vr_group_set_mesh", "vr_group_set_mode", "vr_group_set_seed",
    "vr_group_render", "vr_group_render_device", "vr_group_last_stats", "vr_group_gather_info",
]


class BackendError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vr_backend error {code}: {msg}")
        self.code = code


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.

    libvr_backend.so needs `libamdhip64.so.7`.  The PyTorch wheel ships its own copy of that
    soname; if both it and /opt/rocm's get loaded, the second one to initialise sees no GPU.
    When torch is installed (bench.py and the multi-GPU scheduler use it for RCCL), load its copy
    first so that the dynamic linker resolves our dependency to the same object.
    Set VR_HIP_RUNTIME=system to skip this (stand-alone C/C++ hosts never need it).
    """
    if os.environ.get("VR_HIP_RUNTIME", "") == "system":
        return
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load libvr_backend.so; raises if it has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} is missing: build it with `make -C volume_renderer_amd/csrc` "
                                "(there is no CPU fallback)")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.vr_backend_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.vr_backend_destroy.argtypes = [vp]
    L.vr_backend_destroy.restype = None
    L.vr_backend_last_error.argtypes = [vp]
    L.vr_backend_last_error.restype = C.c_char_p
    L.vr_backend_set_density.argtypes = [vp, C.POINTER(GridDesc), C.POINTER(GridDesc)]
    L.vr_backend_generate_cloud.argtypes = [vp, C.c_int, C.c_float]
    L.vr_backend_read_density.argtypes = [vp, C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.vr_backend_density_max.argtypes = [vp, C.POINTER(C.c_float)]
    L.vr_backend_set_density_map.argtypes = [vp, C.c_float, C.POINTER(C.c_float)]
    L.vr_group_create.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.POINTER(vp)]
    L.vr_group_destroy.argtypes = [vp]
    L.vr_group_destroy.restype = None
    L.vr_group_size.argtypes = [vp]
    L.vr_group_size.restype = C.c_uint32
    L.vr_group_backend.argtypes = [vp, C.c_uint32]
    L.vr_group_backend.restype = vp
    L.vr_group_last_error.argtypes = [vp]
    L.vr_group_last_error.restype = C.c_char_p
    L.vr_group_set_density.argtypes = [vp, C.POINTER(GridDesc), C.POINTER(GridDesc)]
    L.vr_group_generate_cloud.argtypes = [vp, C.c_int, C.c_float]
    L.vr_group_set_density_map.argtypes = [vp, C.c_float, C.POINTER(C.c_float)]
    L.vr_group_set_mesh.argtypes = [vp, C.POINTER(MeshDesc)]
    L.vr_group_set_mode.argtypes = [vp, C.c_int]
    L.vr_group_set_seed.argtypes = [vp, C.c_uint64]
    L.vr_group_render.argtypes = [vp, C.POINTER(OptionsPod), C.c_int, C.POINTER(C.c_float)]
    L.vr_group_render_device.argtypes = [vp, C.POINTER(OptionsPod), C.c_int, vp]
    L.vr_group_last_stats.argtypes = [vp, C.POINTER(RenderStats), C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    L.vr_group_gather_info.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.vr_backend_set_mesh.argtypes = [vp, C.POINTER(MeshDesc)]
    L.vr_backend_set_mode.argtypes = [vp, C.c_int]
    L.vr_backend_set_seed.argtypes = [vp, C.c_uint64]
    L.vr_backend_set_sample_buffer_bytes.argtypes = [vp, C.c_uint64]
    L.vr_backend_render_rows.argtypes = [vp, C.POINTER(OptionsPod), C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
    L.vr_backend_render.argtypes = [vp, C.POINTER(OptionsPod), C.c_int, C.POINTER(C.c_float)]
    L.vr_backend_render_row_set_device.argtypes = [vp, C.POINTER(OptionsPod), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.vr_backend_render_row_set_device_noalloc.argtypes = L.vr_backend_render_row_set_device.argtypes
    L.vr_backend_synchronize.argtypes = [vp]
    L.vr_backend_last_stats.argtypes = [vp, C.POINTER(RenderStats), C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    L.vr_backend_tonemap_device.argtypes = [vp, vp, C.c_uint64, C.c_float, C.POINTER(C.c_float)]
    L.vr_backend_max_channel_device.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_float)]
    L.vr_rgb_to_u8.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)]
    L.vr_abi_sizes.argtypes = [C.POINTER(C.c_uint32)]
    L.vr_abi_sizes.restype = None
    for name in EXPORTS:
        if getattr(L, name).restype is C.c_int:
            pass  # default restype
    sizes = (C.c_uint32 * 4)()
    L.vr_abi_sizes(sizes)
    want = [C.sizeof(OptionsPod), C.sizeof(GridDesc), C.sizeof(MeshDesc), C.sizeof(RenderStats)]
    if list(sizes) != want:
        raise RuntimeError(f"ABI mismatch between abi.py {want} and libvr_backend.so {list(sizes)}")
    _lib = L
    return L


def integrator_id(name_or_id):
    if isinstance(name_or_id, str):
        try:
            return abi.INTEGRATORS[name_or_id]
        except KeyError:
            raise ValueError(f"unknown integrator {name_or_id!r}; the reference's factory knows "
                             f"{sorted(abi.INTEGRATORS)}") from None
    return int(name_or_id)


class Backend:
    """One GPU's integrator backend (vr_backend*)."""

    def __init__(self, device=0):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.vr_backend_create(int(device), C.byref(h))
        if rc != 0:
            raise BackendError(rc, self._L.vr_backend_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._L.vr_backend_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise BackendError(rc, self._L.vr_backend_last_error(self._h).decode())

    # ---- scene ----
    def set_density(self, density, temperature=None):
        t = C.byref(temperature.desc) if temperature is not None else None
        self._check(self._L.vr_backend_set_density(self._h, C.byref(density.desc), t))

    def generate_cloud(self, n, radius_voxels):
        self._check(self._L.vr_backend_generate_cloud(self._h, int(n), float(radius_voxels)))

    def set_density_map(self, voxel_size, origin):
        self._check(self._L.vr_backend_set_density_map(self._h, float(voxel_size), (C.c_float * 3)(*[float(x) for x in origin])))

    def read_density(self):
        dims = (C.c_int32 * 3)()
        bmin = (C.c_int32 * 3)()
        self._check(self._L.vr_backend_read_density(self._h, None, 0, dims, bmin))
        out = np.empty(tuple(dims), dtype=np.float32)
        self._check(self._L.vr_backend_read_density(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size, dims, bmin))
        return out, tuple(bmin)

    def density_max(self):
        v = C.c_float()
        self._check(self._L.vr_backend_density_max(self._h, C.byref(v)))
        return v.value

    def set_mesh(self, mesh):
        self._check(self._L.vr_backend_set_mesh(self._h, C.byref(mesh.desc)))

    # ---- knobs ----
    def set_mode(self, mode):
        self._check(self._L.vr_backend_set_mode(self._h, int(mode)))

    def set_seed(self, seed):
        self._check(self._L.vr_backend_set_seed(self._h, int(seed)))

    def set_sample_buffer_bytes(self, nbytes):
        self._check(self._L.vr_backend_set_sample_buffer_bytes(self._h, int(nbytes)))

    # ---- hot path ----
    def render_rows(self, opts, integrator, from_row, to_row):
        """renderPartial: absolute rows [from_row, to_row) -> float32 (rows, width, 3) on the host."""
        rows = max(0, int(to_row) - int(from_row))
        pix = np.zeros((rows, opts.width, 3), dtype=np.float32)
        self._check(self._L.vr_backend_render_rows(self._h, C.byref(opts), integrator_id(integrator), int(from_row),
                                                   int(to_row), pix.ctypes.data_as(C.POINTER(C.c_float))))
        return pix

    def render(self, opts, integrator):
        """render minus tonemap/save -> float32 (height, width, 3) on the host."""
        fb = np.zeros((opts.height, opts.width, 3), dtype=np.float32)
        self._check(self._L.vr_backend_render(self._h, C.byref(opts), integrator_id(integrator),
                                              fb.ctypes.data_as(C.POINTER(C.c_float))))
        return fb

    def render_row_set_device(self, opts, integrator, row_first, row_stride, row_count, device_ptr):
        """Asynchronous; device_ptr -> row_count*width*3 floats in HBM on this backend's GPU."""
        self._check(self._L.vr_backend_render_row_set_device(self._h, C.byref(opts), integrator_id(integrator),
                                                             int(row_first), int(row_stride), int(row_count),
                                                             C.c_void_p(int(device_ptr))))

    def render_row_set_device_noalloc(self, opts, integrator, row_first, row_stride, row_count, device_ptr):
        """The same frame with every scratch buffer taken from the library's static device array: no allocation."""
        self._check(self._L.vr_backend_render_row_set_device_noalloc(self._h, C.byref(opts), integrator_id(integrator),
                                                                     int(row_first), int(row_stride), int(row_count),
                                                                     C.c_void_p(int(device_ptr))))

    def synchronize(self):
        self._check(self._L.vr_backend_synchronize(self._h))

    def last_stats(self):
        st = RenderStats()
        ms = C.c_float()
        n = C.c_uint32()
        self._check(self._L.vr_backend_last_stats(self._h, C.byref(st), C.byref(ms), C.byref(n)))
        d = st.as_dict()
        d["kernel_ms"] = ms.value
        d["kernel_launches"] = n.value
        return d

    # ---- post ----
    def tonemap_device(self, device_ptr, npixels, max_white=-1.0):
        out = C.c_float()
        self._check(self._L.vr_backend_tonemap_device(self._h, C.c_void_p(int(device_ptr)), int(npixels), float(max_white),
                                                      C.byref(out)))
        return out.value

    def max_channel_device(self, device_ptr, npixels):
        out = C.c_float()
        self._check(self._L.vr_backend_max_channel_device(self._h, C.c_void_p(int(device_ptr)), int(npixels), C.byref(out)))
        return out.value


class BackendGroup:
    """One frame over several GPUs in one process (vr_backend_group*): device i of N renders rows i::N.

    `devices` may name a GPU more than once (a rehearsal of N > 1 on a one-GPU box)."""

    def __init__(self, devices):
        self._L = lib()
        devs = [int(d) for d in devices]
        h = C.c_void_p()
        rc = self._L.vr_group_create((C.c_int * len(devs))(*devs), len(devs), C.byref(h))
        if rc != 0:
            raise BackendError(rc, self._L.vr_group_last_error(None).decode())
        self._h = h
        self.devices = devs

    def close(self):
        if getattr(self, "_h", None):
            self._L.vr_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __len__(self):
        return int(self._L.vr_group_size(self._h))

    def _check(self, rc):
        if rc != 0:
            raise BackendError(rc, self._L.vr_group_last_error(self._h).decode())

    def set_density(self, density, temperature=None):
        t = C.byref(temperature.desc) if temperature is not None else None
        self._check(self._L.vr_group_set_density(self._h, C.byref(density.desc), t))

    def generate_cloud(self, n, radius_voxels):
        self._check(self._L.vr_group_generate_cloud(self._h, int(n), float(radius_voxels)))

    def set_density_map(self, voxel_size, origin):
        self._check(self._L.vr_group_set_density_map(self._h, float(voxel_size), (C.c_float * 3)(*[float(x) for x in origin])))

    def set_mesh(self, mesh):
        self._check(self._L.vr_group_set_mesh(self._h, C.byref(mesh.desc)))

    def set_mode(self, mode):
        self._check(self._L.vr_group_set_mode(self._h, int(mode)))

    def set_seed(self, seed):
        self._check(self._L.vr_group_set_seed(self._h, int(seed)))

    def render(self, opts, integrator, out=None):
        """BaseIntegrator::render minus tonemap/save over all devices -> float32 (height, width, 3) on the host."""
        fb = np.zeros((opts.height, opts.width, 3), dtype=np.float32) if out is None else out
        assert fb.dtype == np.float32 and fb.shape == (opts.height, opts.width, 3) and fb.flags.c_contiguous
        self._check(self._L.vr_group_render(self._h, C.byref(opts), integrator_id(integrator), fb.ctypes.data_as(C.POINTER(C.c_float))))
        return fb

    def gather_info(self):
        """{"peers_enabled", "peers_refused", "frame_path"} of the last gathers (vr_group_gather_info); frame_path: "registered",
        "staged" (pinned staging buffer) or "pageable"."""
        a = (C.c_uint32 * 3)()
        self._check(self._L.vr_group_gather_info(self._h, a))
        return {"peers_enabled": int(a[0]), "peers_refused": int(a[1]), "frame_path": {0: "pageable", 1: "registered", 2: "staged"}.get(int(a[2]), str(int(a[2])))}

    def render_device(self, opts, integrator, device_ptr):
        """The same with the frame gathered at device_ptr (height*width*3 floats) on the group's first GPU."""
        self._check(self._L.vr_group_render_device(self._h, C.byref(opts), integrator_id(integrator), C.c_void_p(int(device_ptr))))

    def tonemap_device(self, device_ptr, npixels, max_white=-1.0):
        out = C.c_float()
        b0 = self._L.vr_group_backend(self._h, 0)
        rc = self._L.vr_backend_tonemap_device(b0, C.c_void_p(int(device_ptr)), int(npixels), float(max_white), C.byref(out))
        if rc != 0:
            raise BackendError(rc, self._L.vr_backend_last_error(b0).decode())
        return out.value

    def last_stats(self):
        st = RenderStats()
        ms = C.c_float()
        n = C.c_uint32()
        self._check(self._L.vr_group_last_stats(self._h, C.byref(st), C.byref(ms), C.byref(n)))
        d = st.as_dict()
        d["kernel_ms"] = ms.value
        d["kernel_launches"] = n.value
        return d


def rgb_to_u8(rgb):
    """saveFile's pixel conversion (BaseIntegrator.cpp:19-25): float (h, w, 3) -> uint8 top-down."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    out = np.zeros((h, w, 3), dtype=np.uint8)
    rc = lib().vr_rgb_to_u8(rgb.ctypes.data_as(C.POINTER(C.c_float)), w, h, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc != 0:
        raise BackendError(rc, "vr_rgb_to_u8")
    return out
