"""The C-ABI library loads without a GPU, exports every symbol include/vr_backend.h declares,
agrees with abi.py on struct layout, and reports (never throws) when no device is present."""
import ctypes as C
import os
import re

import numpy as np

import oracle_lib as ol
from volume_renderer_amd import abi, backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "vr_backend.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vr_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    L = backend.lib()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/vr_backend.h but not exported"
    assert sorted(backend.EXPORTS) == names


def test_struct_sizes_match_the_python_mirror():
    L = backend.lib()
    sizes = (C.c_uint32 * 4)()
    L.vr_abi_sizes(sizes)
    assert list(sizes) == [C.sizeof(abi.OptionsPod), C.sizeof(abi.GridDesc), C.sizeof(abi.MeshDesc),
                           C.sizeof(abi.RenderStats)]


def test_oracle_uses_the_same_structs(oracle):
    o = abi.OptionsPod()
    oracle.vro_options_defaults(C.byref(o))
    d = abi.default_options()
    assert bytes(o) == bytes(d)           # XMLManager.cpp:39-107 defaults, both restatements


def test_create_without_gpu_returns_an_error_code():
    L = backend.lib()
    h = C.c_void_p()
    rc = L.vr_backend_create(0, C.byref(h))
    if rc == 0:                            # on the GPU box
        L.vr_backend_destroy(h)
        return
    assert rc == abi.VR_ERR_NO_DEVICE and not h.value
    assert b"HIP" in L.vr_backend_last_error(None) or L.vr_backend_last_error(None)
    # NULL handles are rejected, not dereferenced
    assert L.vr_backend_set_mode(None, 0) == abi.VR_ERR_INVALID_ARG
    assert L.vr_backend_render(None, None, 0, None) == abi.VR_ERR_INVALID_ARG


def test_unknown_integrator_name_raises():
    import pytest
    with pytest.raises(ValueError):
        backend.integrator_id("pathTracing")        # not a name of the reference's factory
    assert backend.integrator_id("densitySampling") == 7
    assert backend.integrator_id("deltaTracking") == 3
    assert backend.integrator_id("nanoVDBSimple") == 4 and backend.integrator_id("nanoVDBEmission") == 5
    assert backend.integrator_id("heterogeneousPerlinNoise") == 6


def test_rgb_to_u8_matches_oracle_savefile_conversion(oracle):
    rng = np.random.default_rng(1)
    img = rng.uniform(-0.2, 1.3, size=(7, 5, 3)).astype(np.float32)
    img[0, 0] = [0.0, 1.0, 0.999999]
    got = backend.rgb_to_u8(img)
    want = np.zeros_like(got)
    oracle.vro_to_rgb8(ol.fptr(img), 5, 7, want.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert np.array_equal(got, want)
    assert got[0, 0].tolist() == [0, 255, 254]   # truncation, not rounding (BaseIntegrator.cpp:21-23)


def test_host_cpp_mirror_compiles():
    """volume_renderer_amd/host: the C++ mirror of IntegratorFactory / BaseIntegrator over the C-ABI and its
    example are valid C++17 against include/vr_backend.h (syntax and types only: no GPU, no link)."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        import pytest
        pytest.skip("g++ not available")
    src = os.path.join(ROOT, "volume_renderer_amd", "host", "example_render.cpp")
    r = subprocess.run([gxx, "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
