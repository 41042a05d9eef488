"""MI355X-native integrator hot path of emiliano-luna/volume-renderer.

csrc/   HIP kernels + the C-ABI (include/vr_backend.h) -> libvr_backend.so
abi.py  ctypes mirror of include/vr_types.h
backend.py  ctypes binding of the C-ABI
"""
from .abi import (INTEGRATORS, Grid, GridDesc, Mesh, MeshDesc, OptionsPod, RenderStats, copy_options,  # noqa: F401
                  default_options, set_options)
from .backend import (MODE_EXACT, MODE_EXACT_LOCAL, MODE_EXACT_SKIP, MODE_FAST, MODE_FAST_LOCAL, Backend, BackendError, BackendGroup, lib,  # noqa: F401
                      rgb_to_u8)
