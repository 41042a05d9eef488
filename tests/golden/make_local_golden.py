#!/usr/bin/env python3
"""Fixtures for the local-majorant delta tracker (VR_MODE_FAST_LOCAL / VR_MODE_EXACT_LOCAL) from the CPU oracle —
its own restatement of that estimator (oracle/vr_oracle.c, cast_ray_delta_local) and the reference-faithful delta tracker
with the two study switches that separate the two (VRO_DELTA_LOOKUP_AFTER_STEP, VRO_DELTA_CLAMP_ENTRY).  Everything is run
the way the reference runs (mt19937 per row chunk, libm), so the streams are independent of the GPU's.

    python tests/golden/make_local_golden.py            (about 5 minutes on 8 cores)

Writes tests/golden/local_fixtures.npz (data only):
  cloud_<variant>_mean / _stderr   64x64x3 at 4096 spp on the 64^3 cloud, textbook regime (make_stat_golden's frame);
                                   variants: local (cells of 8^3 voxels, the backend's rule for a 64^3 grid), after (lookup
                                   after the step), entry (clamped entry), both; the unswitched tracker is
                                   stat_fixtures.npz's delta_textbook
  box_<voxel>_<variant>            image mean and its standard error (2 floats) of the homogeneous scattering box as
                                   64^3 voxels of size 1, 16^3 of size 4, 8^3 of size 8, at 1024 spp; variants as above + plain
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import make_stat_golden as mg  # noqa: E402
import oracle_lib as ol  # noqa: E402
import scenes  # noqa: E402
from volume_renderer_amd import Grid  # noqa: E402

DELTA = 3
VARIANTS = [("plain", DELTA, 0), ("after", DELTA, ol.DELTA_LOOKUP_AFTER_STEP), ("entry", DELTA, ol.DELTA_CLAMP_ENTRY),
            ("both", DELTA, ol.DELTA_LOOKUP_AFTER_STEP | ol.DELTA_CLAMP_ENTRY), ("local", ol.DELTA_LOCAL, 0)]
BOXES = [("1", lambda: Grid(np.full((64, 64, 64), 0.5, np.float32))),
         ("4", lambda: Grid(np.full((16, 16, 16), 2.0, np.float32), voxel_size=4.0)),
         ("8", lambda: Grid(np.full((8, 8, 8), 4.0, np.float32), voxel_size=8.0))]
BOX_SPP = 1024


def box_options(spp=BOX_SPP):
    return scenes.options_cloud(DELTA, 64, 64, 64, spp, regime="textbook")


def main():
    L = ol.lib()
    L.vro_set_math_mode(ol.MATH_LIBM)
    out = {"spp": np.array(mg.SPP), "box_spp": np.array(BOX_SPP)}
    grid = mg.cloud()
    scene, _keep = ol.make_scene(density=grid)
    ol.local_majorants(scene)
    o = mg.options(DELTA, "textbook")
    for name, integ, variant in VARIANTS[1:]:
        L.vro_set_delta_variant(variant)
        mean, se = mg.pixel_statistics(scene, o, integ, ol.RNG_REFERENCE)
        L.vro_set_delta_variant(0)
        out[f"cloud_{name}_mean"] = mean.astype(np.float32)
        out[f"cloud_{name}_stderr"] = se.astype(np.float32)
        print("cloud", name, "image mean", mean.mean(), flush=True)
    for vox, make in BOXES:
        g = make()
        sc, _k = ol.make_scene(density=g)
        ol.local_majorants(sc)
        for name, integ, variant in VARIANTS:
            L.vro_set_delta_variant(variant)
            mean, _ = mg.pixel_statistics(sc, box_options(), integ, ol.RNG_REFERENCE)
            L.vro_set_delta_variant(0)
            rows = mean.mean(axis=(1, 2))                     # 64 row means, independent streams (one mt19937 per row)
            out[f"box_{vox}_{name}"] = np.array([rows.mean(), rows.std(ddof=1) / np.sqrt(len(rows))])
            print("box", vox, name, out[f"box_{vox}_{name}"], flush=True)
    L.vro_local_majorants_free()
    np.savez_compressed(os.path.join(HERE, "local_fixtures.npz"), **out)


if __name__ == "__main__":
    main()
