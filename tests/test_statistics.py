"""SURVEY 8(c)(iii) on the CPU: the oracle against itself.  tests/golden/stat_fixtures.npz holds per-pixel mean and
standard error at 4096 samples per pixel from the oracle run as the reference runs (mt19937 per row chunk, libm;
tests/golden/make_stat_golden.py).  Here the oracle's OTHER random scheme — the Philox/xoshiro streams the GPU
shares — renders a band of the same frame at the same sample count and must meet the per-pixel bar against the
fixture.  That this passes is what says the bar is fair for the stochastic integrators (heavy tails included: each
side brings its own standard error), so the GPU is held to it unchanged in tests/test_gpu_statistics.py."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol
import stat_common as sc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_stat_golden as mg  # noqa: E402

FIXTURES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stat_fixtures.npz")
BAND = slice(24, 40)          # 16 of the 64 rows, through the cloud's core and both rims: a quarter of the CPU time


@pytest.mark.parametrize("name,integ,regime", mg.CASES)
def test_oracle_philox_streams_meet_the_per_pixel_bar_against_the_reference_scheme(oracle, name, integ, regime):
    fx = np.load(FIXTURES)
    assert int(fx["spp"]) == mg.SPP == 4096
    oracle.vro_set_math_mode(ol.MATH_LIBM)
    scene, _keep = ol.make_scene(density=mg.cloud())
    o = mg.options(integ, regime)

    def row(j):
        _, ps, _ = ol.render_rows(scene, o, integ, j, j + 1, rng_mode=ol.RNG_PHILOX, seed=0xC0FFEE, per_sample=True)
        ps = ps[0].astype(np.float64)
        return ps.mean(axis=1), ps.std(axis=1, ddof=1) / np.sqrt(ps.shape[1])

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as ex:
        rows = list(ex.map(row, range(BAND.start, BAND.stop)))
    mean = np.stack([r[0] for r in rows])
    se = np.stack([r[1] for r in rows])
    r = sc.compare(mean, se, fx[name + "_mean"][BAND], fx[name + "_stderr"][BAND])
    print(name, r)
    assert r["frac_z4"] >= sc.PIXEL_FRACTION, r
    assert r["frac_z3"] > 0.99, r
    # (the image-mean bar is taken over the whole frame on the GPU; a 16-row band of the ratio tracker's
    # NEE spikes is three times noisier)
    assert r["image_mean_rel"] < 3 * sc.IMAGE_MEAN_REL, r


def test_oracle_one_clock_per_walk_meets_the_bar_against_the_reference_scheme(oracle):
    """VRO_RNG_PHILOX_GEOM — what the FAST delta kernel draws where every free path is clamped to stepSizeMin (the shipped
    delta config): ONE exponential clock per walk instead of two numbers per step.  Same distribution of every step's
    outcome (IntegratorDeltaTracking.cpp:62-99), so the same per-pixel bar against the reference-scheme fixture."""
    fx = np.load(FIXTURES)
    oracle.vro_set_math_mode(ol.MATH_LIBM)
    scene, _keep = ol.make_scene(density=mg.cloud())
    o = mg.options(3, "shipped")

    def row(j):
        _, ps, st = ol.render_rows(scene, o, 3, j, j + 1, rng_mode=ol.RNG_PHILOX_GEOM, seed=0xC10C, per_sample=True)
        ps = ps[0].astype(np.float64)
        return ps.mean(axis=1), ps.std(axis=1, ddof=1) / np.sqrt(ps.shape[1])

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as ex:
        rows = list(ex.map(row, range(BAND.start, BAND.stop)))
    mean = np.stack([r[0] for r in rows])
    se = np.stack([r[1] for r in rows])
    r = sc.compare(mean, se, fx["delta_shipped_mean"][BAND], fx["delta_shipped_stderr"][BAND])
    print("delta_shipped, one clock per walk:", r)
    assert r["frac_z4"] >= sc.PIXEL_FRACTION and r["frac_z3"] > 0.99, r
    assert r["image_mean_rel"] < sc.IMAGE_MEAN_REL, r
