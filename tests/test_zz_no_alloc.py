"""The allocation-free entry point (vr_backend_render_row_set_device_noalloc, include/vr_backend.h) against the
entry point it is a variant of (vr_backend_render_row_set_device), through the C-ABI.

No counterpart in the reference.  The bar is the one of every integer/bit-pattern comparison here: the two entry
points share their kernels and their random streams, so frames and work counters must be bit-identical — in every
kernel family, on two calls in a row of which the second is smaller (the static scratch array keeps the first
call's contents: counters, work counters and staging must not leak into the second frame), and on a frame whose
sub-samples do not fit the 128 MiB staging area at once (several passes).
"""
import numpy as np
import pytest

import oracle_lib as ol
import scenes
from volume_renderer_amd import MODE_EXACT_SKIP, MODE_FAST, MODE_FAST_LOCAL, Backend, BackendError, Grid

pytestmark = pytest.mark.gpu

RATIO, DELTA, NEE, IMPROVED, VDB_SIMPLE = 2, 3, 0, 1, 4
COUNTS = ("path_samples", "density_lookups", "scatter_events", "shadow_rays", "ray_queries", "step_limit_hits")


def bitexact(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.fixture(scope="module")
def backend():
    b = Backend(0)
    yield b
    b.close()


@pytest.fixture(scope="module")
def cloud64(oracle):
    n = 64
    vox = np.zeros((n, n, n), dtype=np.float32)
    oracle.vro_make_cloud(n, scenes.cloud_radius(n), ol.fptr(vox), 8)
    return Grid(vox)


def both(backend, o, integ, row_first=0, row_stride=1, rows=None):
    """(frame, counters) of the allocating entry point and of the allocation-free one, same arguments."""
    import torch
    rows = o.height if rows is None else rows
    out = []
    for call in (backend.render_row_set_device, backend.render_row_set_device_noalloc):
        dev = torch.full((rows, o.width, 3), -3.0, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        call(o, integ, row_first, row_stride, rows, dev.data_ptr())
        backend.synchronize()
        st = backend.last_stats()
        out.append((dev.cpu().numpy(), {k: st[k] for k in COUNTS}))
    return out


def test_noalloc_matches_allocating_entry_point_two_calls_in_a_row(backend, cloud64):
    backend.set_density(cloud64)
    backend.set_mesh(scenes.icosphere_mesh())
    backend.set_seed(scenes.SEED)
    # first call large, second smaller (fewer pixels, fewer sub-samples, another row set): per kernel family
    for integ, mode, regime in ((RATIO, MODE_EXACT_SKIP, "shipped"), (DELTA, MODE_FAST, "textbook"), (VDB_SIMPLE, MODE_FAST, "shipped"),
                                (RATIO, MODE_FAST, "shipped"), (DELTA, MODE_FAST, "shipped")):     # (the last: closed-form voxel runs, GEOM)
        backend.set_mode(mode)
        big = scenes.options_cloud(integ, 64, 83, 61, 16, regime=regime)
        small = scenes.options_cloud(integ, 64, 37, 29, 4, regime=regime)
        (a, ast), (b, bst) = both(backend, big, integ)
        assert bitexact(a, b) and ast == bst, (integ, "big")
        assert ast["path_samples"] == 83 * 61 * 16
        assert np.all(a != -3.0)
        (a, ast), (b, bst) = both(backend, small, integ, row_first=1, row_stride=2, rows=14)
        assert bitexact(a, b) and ast == bst, (integ, "small")
        assert ast["path_samples"] == 37 * 14 * 4
    # the kernels added in round 2: the local-majorant delta tracker and the mesh-bounded trackers
    backend.set_mode(MODE_FAST_LOCAL)
    (a, ast), (b, bst) = both(backend, scenes.options_cloud(DELTA, 64, 83, 61, 16, regime="textbook"), DELTA)
    assert bitexact(a, b) and ast == bst and ast["path_samples"] == 83 * 61 * 16
    (a, ast), (b, bst) = both(backend, scenes.options_cloud(DELTA, 64, 37, 29, 4, regime="textbook"), DELTA, row_first=1, row_stride=2, rows=14)
    assert bitexact(a, b) and ast == bst
    backend.set_density_map(*scenes.mesh_tracker_grid_map(64))
    for integ in (8, 9):                                             # meshRatioTracking, meshDeltaTracking
        backend.set_mode(MODE_EXACT_SKIP)
        (a, ast), (b, bst) = both(backend, scenes.options_mesh_tracker(integ, 64, 70, 50, 4), integ)
        assert bitexact(a, b) and ast == bst and ast["ray_queries"] >= 70 * 50 * 4, (integ, "big")
        (a, ast), (b, bst) = both(backend, scenes.options_mesh_tracker(integ, 64, 31, 23, 1), integ)
        assert bitexact(a, b) and ast == bst, (integ, "small")
    backend.set_density(cloud64)
    for integ in (NEE, IMPROVED):
        backend.set_mode(MODE_EXACT_SKIP)
        (a, ast), (b, bst) = both(backend, scenes.options_marcher(90, 70, 4), integ)
        assert bitexact(a, b) and ast == bst, (integ, "big")
        (a, ast), (b, bst) = both(backend, scenes.options_marcher(33, 21, 1), integ)
        assert bitexact(a, b) and ast == bst, (integ, "small")


def test_noalloc_allocates_nothing(cloud64):
    """What the name promises, measured: a fresh backend whose allocating path has never run renders through the
    allocation-free entry point — tracking, local-majorant and marching kernels — and the device's free memory does not move
    (hipMemGetInfo before and after; the allocating entry point, run afterwards on the same backend, does take memory)."""
    import torch
    with Backend(0) as b:
        b.set_density(cloud64)
        b.set_mesh(scenes.icosphere_mesh())
        b.set_seed(scenes.SEED)
        dev = torch.zeros((61, 83, 3), dtype=torch.float32, device="cuda:0")
        big = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        for integ, mode, o in ((RATIO, MODE_FAST, scenes.options_cloud(RATIO, 64, 83, 61, 16)),
                               (DELTA, MODE_FAST_LOCAL, scenes.options_cloud(DELTA, 64, 83, 61, 16, regime="textbook")),
                               (NEE, MODE_FAST, scenes.options_marcher(83, 61, 4))):
            b.set_mode(mode)
            b.render_row_set_device_noalloc(o, integ, 0, 1, 61, dev.data_ptr())
            b.synchronize()
            assert b.last_stats()["path_samples"] == 83 * 61 * o.rayPerPixelCount
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0
        # the same frame size through both: 1080p at 16 spp needs 398 MB of staging, which the allocating entry point takes from
        # the device and the allocation-free one renders in passes through its static array
        o = scenes.options_cloud(DELTA, 64, 1920, 1080, 16, regime="textbook")
        b.set_mode(MODE_FAST)
        b.render_row_set_device_noalloc(o, DELTA, 0, 1, 1080, big.data_ptr())
        b.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0
        b.render_row_set_device(o, DELTA, 0, 1, 1080, big.data_ptr())
        b.synchronize()
        assert torch.cuda.mem_get_info()[0] <= free0 - 300 * 2**20   # the staging buffer of the allocating path


def test_noalloc_renders_in_passes_when_the_staging_area_is_short(backend, cloud64):
    """1920 x 1080 at 16 spp needs 398 MB of staging: the 128 MiB area takes it in 5-sample passes (the
    allocating entry point stages all 16 at once)."""
    backend.set_density(cloud64)
    backend.set_seed(scenes.SEED)
    backend.set_mode(MODE_FAST)
    o = scenes.options_cloud(DELTA, 64, 1920, 1080, 16, regime="textbook")
    (a, ast), (b, bst) = both(backend, o, DELTA)
    assert bitexact(a, b) and ast == bst
    assert backend.last_stats()["kernel_launches"] == 4       # ceil(16 / 5) passes of at most 5 sub-samples
    assert backend.last_stats()["kernel_ms"] == 0.0           # no events: no time measured


def test_noalloc_rejects_what_it_cannot_hold(backend, cloud64):
    import torch
    backend.set_density(cloud64)
    dev = torch.zeros(16, dtype=torch.float32, device="cuda:0")
    o = scenes.options_cloud(DELTA, 64, 4096, 2731, 1, regime="textbook")      # 11 186 176 pixels > 11 184 810
    with pytest.raises(BackendError, match="allocation-free"):
        backend.render_row_set_device_noalloc(o, DELTA, 0, 1, 2731, dev.data_ptr())
    o = scenes.options_cloud(DELTA, 64, 4, 4, 1 << 17, regime="textbook")
    with pytest.raises(BackendError, match="allocation-free"):
        backend.render_row_set_device_noalloc(o, DELTA, 0, 1, 4, dev.data_ptr())
    # the allocating entry point still takes the second one's sub-sample count
    o = scenes.options_cloud(DELTA, 64, 4, 4, 4, regime="textbook")
    dev = torch.zeros((4, 4, 3), dtype=torch.float32, device="cuda:0")
    backend.render_row_set_device_noalloc(o, DELTA, 0, 1, 4, dev.data_ptr())
    backend.synchronize()
