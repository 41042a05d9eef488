"""VR_MODE_FAST_LOCAL / VR_MODE_EXACT_LOCAL: delta tracking over per-cell majorants (a coarse majorant grid walked by a 3-D
DDA) — north_star "DensityFields".  Opt-in and NOT the reference's estimator (vr_kernels.h, vr_delta_dda_kernel): it is
textbook delta tracking.  Its oracle is the CPU restatement of the same estimator (oracle/vr_oracle.c,
cast_ray_delta_local): the EXACT build of the kernel must equal it bit for bit (image and counters), the production
(FAST) build must meet SURVEY 8(c)(iii)'s per-pixel bar against that oracle's 4096-spp fixture
(tests/golden/local_fixtures.npz; how the estimator relates to the reference's is measured on the CPU in
tests/test_local_majorant_oracle.py).  Then the properties textbook delta tracking has and the reference's tracker partly
lacks — exp(-tau) through any slab (layered ones included), independence of the voxelisation — and the refusal to run
where the reference's parameters make its tracker something else."""
import math

import numpy as np
import pytest

import oracle_lib as ol
import scenes
import os
import sys

import stat_common as sc
from volume_renderer_amd import (MODE_EXACT_LOCAL, MODE_FAST, MODE_FAST_LOCAL, Backend, BackendError, BackendGroup, Grid,
                                 copy_options, default_options)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_stat_golden as mg  # noqa: E402

LOCAL_FIXTURES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "local_fixtures.npz")

pytestmark = pytest.mark.gpu
RATIO, DELTA = 2, 3


@pytest.fixture(scope="module")
def backend():
    b = Backend(0)
    yield b
    b.close()


def bitexact(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def oracle_local(oracle, scene, o, cell_log):
    """The oracle's restatement of the estimator in the arithmetic and streams the EXACT kernel shares."""
    ol.local_majorants(scene, cell_log)
    oracle.vro_set_math_mode(ol.MATH_SHARED)
    try:
        return ol.render(scene, o, ol.DELTA_LOCAL, rng_mode=ol.RNG_PHILOX, seed=scenes.SEED, want_stats=True)
    finally:
        oracle.vro_set_math_mode(ol.MATH_LIBM)
        oracle.vro_local_majorants_free()


def _ragged_grid():
    """Dims that are no multiples of a brick or a cell (cells stick out of the bbox), a positive background in the
    padding, holes of zeros (empty cells) and a few negative voxels."""
    rng = np.random.default_rng(23)
    n = (37, 50, 29)
    x, y, z = np.meshgrid(*(np.arange(m, dtype=np.float32) for m in n), indexing="ij")
    blob = np.sin(x * 0.31) * np.cos(y * 0.23) * np.sin(z * 0.41 + 1.0)
    vox = np.where(blob > 0.15, blob, 0.0).astype(np.float32)
    vox[rng.uniform(size=n) < 0.01] = -0.5
    vox[8:32, 16:48, 0:29] = 0.0                                       # whole 8^3 and 16^3 cells without medium
    return Grid(vox, bbox_min=(-18, -25, -14), background=0.05)


@pytest.mark.parametrize("case", ["cloud_cells8", "cloud_cells32", "ragged_cells8", "ragged_cells16", "rotated_crop_temperature"])
def test_exact_local_is_bit_identical_to_the_oracle_restatement(backend, oracle, monkeypatch, case):
    """vr_delta_dda_kernel<.., EXACT> against cast_ray_delta_local: same Philox/xoshiro streams (one per path), same
    operation order, IEEE divisions and the shared polynomial log / sin / cos — image, tentative collisions (density
    lookups) and scatter events must be equal bit for bit, for every cell size and on grids whose cells stick out of the
    bbox."""
    n = 64
    temp = None
    kw = {}
    if case.startswith("cloud") or case == "rotated_crop_temperature":
        vox = np.zeros((n, n, n), dtype=np.float32)
        oracle.vro_make_cloud(n, scenes.cloud_radius(n), ol.fptr(vox), 8)
        grid = Grid(vox)
        level = 2 if case == "cloud_cells32" else 0
        if case == "rotated_crop_temperature":
            rng = np.random.default_rng(5)
            temp = Grid((rng.uniform(0, 2, size=(40, 40, 40)) * (rng.uniform(size=(40, 40, 40)) < 0.5)).astype(np.float32), bbox_min=(-20, -20, -20))
            kw = dict(widthReference=96, heightReference=72, widthStartOffset=30, heightStartOffset=20, cameraPosition=(20.0, 35.0, 60.0),
                      cameraRotation=(-25.0, 18.0, 3.0), emissionColor=(0.9, 0.4, 0.1), mediumColor=(0.2, 0.3, 0.4),
                      lightRayDensityMultiplier=1.5, heyneyGreensteinG=0.0)
            level = 1
    else:
        grid = _ragged_grid()
        level = 1 if case == "ragged_cells16" else 0
        kw = dict(cameraPosition=(10.0, -6.0, 60.0))
    o = scenes.options_cloud(DELTA, 64, 48, 40, 4, regime="textbook", **kw)
    scene, _keep = ol.make_scene(density=grid, temperature=temp)
    want, wst = oracle_local(oracle, scene, o, 3 + level)
    monkeypatch.setenv("VR_TUNE_MAJORANT_LEVEL", str(level))
    backend.set_density(grid, temp)
    backend.set_mode(MODE_EXACT_LOCAL)
    backend.set_seed(scenes.SEED)
    got = backend.render(o, DELTA)
    st = backend.last_stats()
    assert bitexact(got, want), f"{np.count_nonzero(got.view(np.uint32) != want.view(np.uint32))} of {got.size} channels differ"
    for k in ("path_samples", "density_lookups", "scatter_events"):
        assert st[k] == wst[k], (k, st[k], wst[k])
    assert st["step_limit_hits"] == 0 and wst["scatter_events"] > 100
    assert np.count_nonzero(np.any(want != np.array(o.backgroundColor[:]), axis=-1)) > 50
    # the production build walks the same streams: its frame is the oracle's up to rounding
    backend.set_mode(MODE_FAST_LOCAL)
    fast = backend.render(o, DELTA)
    close = np.isclose(fast, want, rtol=1e-4, atol=1e-7).all(axis=2).mean()
    print(case, "FAST_LOCAL pixels within 1e-4 of the oracle:", close)
    assert close > 0.97, close


def test_fast_local_meets_the_per_pixel_bar_against_its_oracle(backend):
    """SURVEY 8(c)(iii) for the local-majorant tracker: 64 frames of 4096 spp with different keys against the oracle's
    per-pixel mean at 4096 spp, run as the reference runs (mt19937 per row, libm; tests/golden/make_local_golden.py):
    |z| < 4 on >= 99.9 % of the pixel-channels, image mean within 0.5 %, same spread per pixel."""
    fx = np.load(LOCAL_FIXTURES)
    assert int(fx["spp"]) == 4096
    backend.set_density(mg.cloud())
    backend.set_mode(MODE_FAST_LOCAL)
    o = mg.options(DELTA, "textbook")
    frames = []
    for k in range(64):
        backend.set_seed(0xFACE0000 + 977 * k)
        frames.append(backend.render(o, DELTA).astype(np.float64))
    r = sc.compare_frames(np.stack(frames), fx["cloud_local_mean"], fx["cloud_local_stderr"])
    print("FAST_LOCAL against the local-majorant oracle:", r)
    assert r["image_mean_rel"] <= sc.IMAGE_MEAN_REL, r
    assert r["frac_z4"] >= sc.PIXEL_FRACTION and r["frac_z3"] > 0.98, r
    assert 0.9 < r["median_se_ratio"] < 1.1, r


def slab_options(spp, **kw):
    base = dict(width=48, height=48, widthReference=4800, heightReference=4800, widthStartOffset=2376, heightStartOffset=2376,
                rayPerPixelCount=spp, maxDepth=8, sigma_a=0.25, sigma_s=0.0, stepSizeMin=1e-4, stepSizeMax=1e30,
                stepSizeMultiplier=1.0, cameraPosition=(0, 0, 40.0), fov=30.0, backgroundColor=(1.0, 1.0, 1.0),
                mediumColor=(0.0, 0.0, 0.0))
    base.update(kw)
    return default_options(**base)


@pytest.mark.parametrize("level", ["0", "2", "4"])
@pytest.mark.parametrize("case", ["homogeneous", "two_layers", "three_layers_fine", "thick", "hole"])
def test_escape_probability_is_exp_minus_tau_for_any_slab(backend, monkeypatch, case, level):
    """Absorbing slabs, rays parallel to z to within 1e-3 rad, 2.4 M paths: P(escape) = exp(-sigma_a * integral of rho),
    whatever the layering and the size of the majorant cells (8^3, 32^3, 128^3 voxels on a 64^3 grid).  The reference's
    tracker gives 0.134 instead of 0.088 on `two_layers` (tests/test_gpu_statistics.py)."""
    n = 64
    vox = np.zeros((n, n, n), np.float32)
    if case == "homogeneous":
        vox[:] = 0.1; sigma_a = 0.25
    elif case == "two_layers":
        vox[:] = 0.05; vox[:, :, 20:48] = 0.225; sigma_a = 0.3
    elif case == "three_layers_fine":
        vox[:] = 0.02; vox[:, :, 9:11] = 3.0; vox[:, :, 40:41] = 5.0; sigma_a = 0.1
    elif case == "thick":
        vox[:] = 0.25; sigma_a = 0.5
    else:
        vox[:] = 0.2; vox[:, :, 16:40] = 0.0; sigma_a = 0.3       # an empty gap: cells without medium are crossed in one step
    tau = sigma_a * float(vox[0, 0, :].astype(np.float64).sum())
    monkeypatch.setenv("VR_TUNE_MAJORANT_LEVEL", level)
    backend.set_density(Grid(vox))
    backend.set_mode(MODE_FAST_LOCAL)
    backend.set_seed(7)
    o = slab_options(1024, sigma_a=sigma_a, cameraPosition=(0, 0, 160.0), fov=30.0)
    img = backend.render(o, DELTA)
    paths = 48 * 48 * 1024
    want = math.exp(-tau)
    got = float(img[..., 0].astype(np.float64).mean())
    sigma = math.sqrt(want * (1 - want) / paths)
    print(case, level, "escape probability", got, "exp(-tau)", want, "z", (got - want) / sigma)
    assert abs(got - want) < 4 * sigma, (got, want, sigma)


def _frames(backend, grid, o, mode, k, seed0):
    backend.set_density(grid)
    backend.set_mode(mode)
    out = []
    for i in range(k):
        backend.set_seed(seed0 + 31 * i)
        out.append(backend.render(o, DELTA).astype(np.float64))
    return np.stack(out)


def test_image_does_not_depend_on_the_voxelisation(backend):
    """The same homogeneous scattering box as 64^3 voxels of size 1, 16^3 of size 4 and 8^3 of size 8: the local-majorant
    tracker renders the same image (it samples the medium, not the grid).  The reference's tracker does not — a ray that
    enters through a max face starts one voxel outside the grid and crosses its first 0.01 voxel in empty-space steps —
    and the test records the size of that effect (0.2 % / 0.8 % / 1.5 % of the image mean) so that the difference between
    the two modes on coarse grids is not mistaken for an error of either."""
    o = scenes.options_cloud(DELTA, 64, 64, 64, 1024, regime="textbook")
    grids = [Grid(np.full((64, 64, 64), 0.5, np.float32)), Grid(np.full((16, 16, 16), 2.0, np.float32), voxel_size=4.0),
             Grid(np.full((8, 8, 8), 4.0, np.float32), voxel_size=8.0)]
    means, ses, fast = [], [], []
    for g in grids:
        fr = _frames(backend, g, o, MODE_FAST_LOCAL, 8, 0x123000)
        per = fr.mean(axis=(1, 2, 3))
        means.append(per.mean()); ses.append(per.std(ddof=1) / math.sqrt(len(per)))
        fast.append(_frames(backend, g, o, MODE_FAST, 4, 0xABC000).mean())
    print("local", means, "fast", fast)
    for i in (1, 2):
        assert abs(means[i] - means[0]) < 4 * math.hypot(ses[i], ses[0]) + 1e-4 * means[0], (means, ses)
    assert abs(fast[0] - means[0]) / means[0] < 0.004
    assert 0.008 < (means[2] - fast[2]) / means[2] < 0.025          # the reference's estimator on 8-unit voxels


def test_close_to_the_global_majorant_kernel_on_the_cloud_and_far_fewer_collisions(backend, oracle):
    """The 64^3 cloud (its finest features are two voxels wide, a free path of the global majorant is one and a half):
    the two estimators differ by the reference's stale density lookup, 1.5 % of the image mean here (1.0 % on the 512^3
    cloud, 0.8 % on 1024^3); pixel by pixel the frames are the same picture."""
    n = 64
    vox = np.zeros((n, n, n), dtype=np.float32)
    oracle.vro_make_cloud(n, scenes.cloud_radius(n), ol.fptr(vox), 8)
    grid = Grid(vox)
    o = scenes.options_cloud(DELTA, 64, 64, 64, 1024, regime="textbook")
    fast = _frames(backend, grid, o, MODE_FAST, 4, 0xABC000)
    fast_loads = backend.last_stats()["voxel_loads"]
    local = _frames(backend, grid, o, MODE_FAST_LOCAL, 4, 0x123000)
    st = backend.last_stats()
    rel = (local.mean() - fast.mean()) / fast.mean()
    corr = np.corrcoef(local.mean(axis=0).ravel(), fast.mean(axis=0).ravel())[0, 1]
    print("image mean rel diff", rel, "correlation", corr, "collisions per sample", st["density_lookups"] / st["path_samples"],
          "loads per sample of the global-majorant kernel", fast_loads / st["path_samples"])
    assert 0.0 < rel < 0.025 and corr > 0.995
    assert st["density_lookups"] == st["voxel_loads"] and st["voxel_loads"] * 3 < fast_loads
    assert st["step_limit_hits"] == 0 and st["path_samples"] == 64 * 64 * 1024


def test_deterministic_partition_invariant_and_the_other_integrators_untouched(backend, oracle):
    import torch
    n = 64
    vox = np.zeros((n, n, n), dtype=np.float32)
    oracle.vro_make_cloud(n, scenes.cloud_radius(n), ol.fptr(vox), 8)
    grid = Grid(vox)
    backend.set_density(grid)
    backend.set_seed(scenes.SEED)
    o = scenes.options_cloud(DELTA, 64, 70, 37, 8, regime="textbook")
    backend.set_mode(MODE_FAST_LOCAL)
    full = backend.render(o, DELTA)
    assert np.array_equal(full, backend.render(o, DELTA))
    parts = np.zeros_like(full)
    for r in range(3):
        cnt = len(range(r, 37, 3))
        dev = torch.zeros((cnt, 70, 3), dtype=torch.float32, device="cuda:0")
        backend.render_row_set_device(o, DELTA, r, 3, cnt, dev.data_ptr())
        backend.synchronize()
        parts[r::3] = dev.cpu().numpy()
    assert np.array_equal(full, parts)
    with BackendGroup([0, 0]) as g:
        g.set_density(grid); g.set_seed(scenes.SEED); g.set_mode(MODE_FAST_LOCAL)
        assert np.array_equal(full, g.render(o, DELTA))
    # ratio tracking has no local-majorant kernel: the mode renders it exactly as FAST does
    orat = scenes.options_cloud(RATIO, 64, 48, 40, 4, regime="textbook")
    a = backend.render(orat, RATIO)
    backend.set_mode(MODE_FAST)
    assert np.array_equal(a, backend.render(orat, RATIO))


def test_refuses_the_regimes_in_which_the_reference_is_not_a_plain_delta_tracker(backend, oracle):
    n = 16
    backend.set_density(Grid(np.full((n, n, n), 0.3, np.float32)))
    backend.set_mode(MODE_FAST_LOCAL)
    ok = scenes.options_cloud(DELTA, 64, 8, 8, 1, regime="textbook")
    assert backend.render(ok, DELTA).shape == (8, 8, 3)
    for bad in (scenes.options_cloud(DELTA, 64, 8, 8, 1, regime="shipped"),          # stepSizeMultiplier 1e-4, clamps in reach
                copy_options(ok, stepSizeMin=0.5), copy_options(ok, stepSizeMax=10.0), copy_options(ok, useImportanceSampling=1),
                copy_options(ok, lightRayDensityMultiplier=-1.0)):
        with pytest.raises(BackendError) as e:
            backend.render(bad, DELTA)
        assert e.value.code == -6                                                     # VR_ERR_UNSUPPORTED
    backend.set_mode(MODE_FAST)
    assert backend.render(scenes.options_cloud(DELTA, 64, 8, 8, 1, regime="shipped"), DELTA).shape == (8, 8, 3)
