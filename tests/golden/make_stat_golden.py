#!/usr/bin/env python3
"""Per-pixel mean and standard error of the stochastic integrators at 4096 samples per pixel, from the CPU oracle run
the way the reference runs (VRO_RNG_REFERENCE: mt19937 seeded per row chunk, uniform_real_distribution; libm) —
SURVEY 8(c)(iii).  Writes tests/golden/stat_fixtures.npz; tests/test_statistics.py (oracle against itself, CPU) and
tests/test_gpu_statistics.py (the HIP path in FAST mode) compare against it.

    python tests/golden/make_stat_golden.py            (about 6 minutes on 8 cores)
    python tests/golden/make_stat_golden.py --ratio-deep [threads]
        the two ratio-tracking cases again at 65 536 spp -> tests/golden/stat_fixtures_ratio_deep.npz (about an hour on
        6 cores).  A ratio-tracking sample is heavy-tailed (next-event estimation through g = 0.924 under a light of 1024):
        4096 of them are too few for the per-pixel mean to be normal, 65 536 are enough for SURVEY 8(c)(iii)'s bar to be
        asserted as written (tests/test_gpu_statistics.py).

Data only: 64x64 pixels x 3 channels x (mean, stderr) x 4 cases, float32.
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as ol  # noqa: E402
import scenes  # noqa: E402
from volume_renderer_amd import Grid  # noqa: E402

N, W, H, SPP = 64, 64, 64, 4096
CASES = [("ratio_shipped", 2, "shipped"), ("ratio_textbook", 2, "textbook"), ("delta_shipped", 3, "shipped"),
         ("delta_textbook", 3, "textbook")]


def cloud():
    vox = np.zeros((N, N, N), dtype=np.float32)
    ol.lib().vro_make_cloud(N, scenes.cloud_radius(N), ol.fptr(vox), 8)
    return Grid(vox)


def options(integ, regime, spp=SPP):
    return scenes.options_cloud(integ, N, W, H, spp, regime=regime)


def pixel_statistics(scene, o, integ, rng_mode, seed=scenes.SEED, threads=8):
    """(mean, stderr) per pixel and channel, float64, from the per-sample values of every row."""
    def row(j):
        _, ps, _ = ol.render_rows(scene, o, integ, j, j + 1, rng_mode=rng_mode, seed=seed, per_sample=True)
        ps = ps[0].astype(np.float64)                        # (W, spp, 3)
        return ps.mean(axis=1), ps.std(axis=1, ddof=1) / np.sqrt(ps.shape[1])
    with ThreadPoolExecutor(threads) as ex:
        rows = list(ex.map(row, range(o.height)))
    return np.stack([r[0] for r in rows]), np.stack([r[1] for r in rows])


def main():
    ol.lib().vro_set_math_mode(ol.MATH_LIBM)
    grid = cloud()
    scene, _keep = ol.make_scene(density=grid)
    out = {"spp": np.array(SPP), "n": np.array(N)}
    for name, integ, regime in CASES:
        mean, se = pixel_statistics(scene, options(integ, regime), integ, ol.RNG_REFERENCE)
        out[name + "_mean"] = mean.astype(np.float32)
        out[name + "_stderr"] = se.astype(np.float32)
        print(name, "image mean", mean.mean(), "median relative stderr", float(np.median(se / np.maximum(mean, 1e-9))), flush=True)
    np.savez_compressed(os.path.join(HERE, "stat_fixtures.npz"), **out)


DEEP_SPP = 65536
DEEP_CASES = [c for c in CASES if c[1] == 2]


def main_ratio_deep(threads):
    ol.lib().vro_set_math_mode(ol.MATH_LIBM)
    grid = cloud()
    scene, _keep = ol.make_scene(density=grid)
    out = {"spp": np.array(DEEP_SPP), "n": np.array(N)}
    for name, integ, regime in DEEP_CASES:
        mean, se = pixel_statistics(scene, options(integ, regime, DEEP_SPP), integ, ol.RNG_REFERENCE, threads=threads)
        out[name + "_mean"] = mean.astype(np.float32)
        out[name + "_stderr"] = se.astype(np.float32)
        print(name, "image mean", mean.mean(), "median relative stderr", float(np.median(se / np.maximum(mean, 1e-9))), flush=True)
        np.savez_compressed(os.path.join(HERE, "stat_fixtures_ratio_deep.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--ratio-deep":
        main_ratio_deep(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
    else:
        main()
