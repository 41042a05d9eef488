"""One frame over several GPUs behind the C-ABI (vr_group_*, include/vr_backend.h): the fan-out the reference
does inside BaseIntegrator::render (BaseIntegrator.cpp:143-181).  The test box has one GPU, so groups list it more
than once — the row split, the gather and the statistics are the code an 8-GPU node runs; only the peer copies
between different GPUs are not walked here.  The frame must be the one a single vr_backend renders, bit for bit,
for every device count."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
import scenes
from volume_renderer_amd import INTEGRATORS, MODE_EXACT_SKIP, MODE_FAST, Backend, BackendError, BackendGroup, Grid

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "volume_renderer_amd", "host", "example_render")


def bitexact(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.fixture(scope="module")
def cloud(oracle):
    n = 64
    vox = np.zeros((n, n, n), dtype=np.float32)
    oracle.vro_make_cloud(n, scenes.cloud_radius(n), ol.fptr(vox), 8)
    return Grid(vox)


@pytest.mark.parametrize("ndev", [1, 2, 3, 8])
@pytest.mark.parametrize("integ", ["ratioTracking", "deltaTracking"])
def test_group_frame_is_the_single_gpu_frame(cloud, oracle, ndev, integ):
    o = scenes.options_cloud(integ, 64, 70, 37, 4)            # ragged: 37 rows over 2, 3 and 8 devices
    with Backend(0) as one:
        one.set_seed(scenes.SEED)
        one.set_mode(MODE_EXACT_SKIP)
        one.set_density(cloud)
        want = one.render(o, integ)
        wst = one.last_stats()
    sc, _keep = ol.make_scene(density=cloud)
    oracle.vro_set_math_mode(ol.MATH_SHARED)
    try:
        ref = ol.render(sc, o, INTEGRATORS[integ], rng_mode=ol.RNG_PHILOX, seed=scenes.SEED)
    finally:
        oracle.vro_set_math_mode(ol.MATH_LIBM)
    assert bitexact(want, ref)
    with BackendGroup([0] * ndev) as g:
        assert len(g) == ndev
        g.set_seed(scenes.SEED)
        g.set_mode(MODE_EXACT_SKIP)
        g.set_density(cloud)
        got = g.render(o, integ)
        st = g.last_stats()
        assert bitexact(got, ref)
        for k in ("path_samples", "density_lookups", "scatter_events", "shadow_rays"):
            assert st[k] == wst[k], k
        assert st["kernel_launches"] == min(ndev, 37) and st["kernel_ms"] > 0
        # a second frame on the same group (buffers reused), fewer rows than devices
        o2 = scenes.options_cloud(integ, 64, 33, 5, 2)
        got2 = g.render(o2, integ)
    with Backend(0) as one:
        one.set_seed(scenes.SEED)
        one.set_mode(MODE_EXACT_SKIP)
        one.set_density(cloud)
        assert bitexact(got2, one.render(o2, integ))


def test_group_gathers_in_hbm_and_tonemaps_there(cloud):
    import torch
    o = scenes.options_cloud("ratioTracking", 64, 96, 50, 4)
    with BackendGroup([0, 0, 0]) as g:
        g.set_seed(scenes.SEED)
        g.set_mode(MODE_FAST)
        g.set_density(cloud)
        host = g.render(o, "ratioTracking")
        frame = torch.full((50, 96, 3), -1.0, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        g.render_device(o, "ratioTracking", frame.data_ptr())
        assert bitexact(frame.cpu().numpy(), host)
        white = g.tonemap_device(frame.data_ptr(), 50 * 96)
        assert white == pytest.approx(float(host.max()))
        want = host * (host / np.float32(white * white) + 1.0) / (host + 1.0)
        assert np.allclose(frame.cpu().numpy(), want, rtol=1e-6)


def test_group_mesh_scene_and_generated_cloud():
    mesh = scenes.icosphere_mesh()
    o = scenes.options_mesh_tracker("meshDeltaTracking", 64, 64, 45, 4)
    vs, origin = scenes.mesh_tracker_grid_map(64)
    with Backend(0) as one:
        one.set_seed(scenes.SEED)
        one.generate_cloud(64, scenes.cloud_radius(64))
        one.set_density_map(vs, origin)
        one.set_mesh(mesh)
        want = one.render(o, "meshDeltaTracking")
        marcher = one.render(scenes.options_marcher(64, 45, 2), "homogeneousRayMarcherImproved")
        plain = one.render(o, "deltaTracking")
    assert not bitexact(want, plain)
    with BackendGroup([0, 0]) as g:
        g.set_seed(scenes.SEED)
        g.generate_cloud(64, scenes.cloud_radius(64))
        g.set_density_map(vs, origin)
        g.set_mesh(mesh)
        assert bitexact(g.render(o, "meshDeltaTracking"), want)
        assert bitexact(g.render(scenes.options_marcher(64, 45, 2), "homogeneousRayMarcherImproved"), marcher)


def test_group_errors_name_the_device(cloud):
    with BackendGroup([0, 0]) as g:
        o = scenes.options_cloud("ratioTracking", 64, 16, 16, 1)
        with pytest.raises(BackendError) as e:
            g.render(o, "ratioTracking")
        assert e.value.code == -4 and "device 0 of the group" in str(e.value)      # VR_ERR_NO_DENSITY
        g.set_density(cloud)
        with pytest.raises(BackendError) as e:
            g.render(o, "homogeneousRayMarcherNEE")
        assert e.value.code == -5                                                   # VR_ERR_NO_MESH
        assert g.render(o, "ratioTracking").shape == (16, 16, 3)
    with pytest.raises(BackendError) as e:
        BackendGroup([0, 99])
    assert e.value.code == -1 and "device 1 of the list" in str(e.value)
    with pytest.raises(BackendError):
        BackendGroup([])


def test_cpp_host_renders_the_same_frame_on_one_and_on_two_devices():
    """host/example_render.cpp (IntegratorFactory / BaseIntegrator::render over vr_group_render), no Python in the process."""
    if not os.path.exists(EXE):
        pytest.fail("volume_renderer_amd/host/example_render not built: run __graft_entry__.build()")
    out = {}
    for devs in (["0"], ["0", "0"], ["0", "0", "0", "0", "0"]):
        r = subprocess.run([EXE] + devs, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert int(re.search(r"devices (\d+)", r.stdout).group(1)) == len(devs)
        out[len(devs)] = re.search(r"frame hash ([0-9a-f]{16})", r.stdout).group(1)
    assert out[1] == out[2] == out[5], out
