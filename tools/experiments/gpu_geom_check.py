#!/usr/bin/env python3
"""GPU-box check of the GEOM kernel (closed-form voxel runs of the FAST delta tracker where every free path is clamped):
frame against the oracle's step-by-step restatement of the same clock (VRO_RNG_PHILOX_GEOM)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol, scenes
from volume_renderer_amd import MODE_FAST, MODE_EXACT_SKIP, Backend, Grid
L = ol.lib()
n = 64
vox = np.zeros((n, n, n), np.float32); L.vro_make_cloud(n, scenes.cloud_radius(n), ol.fptr(vox), 8)
g = Grid(vox); sc, _k = ol.make_scene(density=g)
o = scenes.options_cloud(3, 64, 64, 64, 16, regime="shipped")
L.vro_set_math_mode(ol.MATH_LIBM)
want, wst = ol.render(sc, o, 3, rng_mode=ol.RNG_PHILOX_GEOM, seed=scenes.SEED, want_stats=True)
plain = ol.render(sc, o, 3, rng_mode=ol.RNG_PHILOX, seed=scenes.SEED)
b = Backend(0); b.set_density(g); b.set_seed(scenes.SEED); b.set_mode(MODE_FAST)
got = b.render(o, 3); st = b.last_stats()
rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
print("within 1e-4 of the GEOM oracle:", np.mean(np.all(rel < 1e-4, axis=-1)), "means", got.mean(), want.mean(), plain.mean())
print("lookups gpu/oracle", st["density_lookups"], wst["density_lookups"], "scatters", st["scatter_events"], wst["scatter_events"], "trips", st["wave_trips"])
os.environ["VR_TUNE_NO_GEOM"] = "1"
got2 = b.render(o, 3); st2 = b.last_stats()
rel2 = np.abs(got2 - plain) / np.maximum(np.abs(plain), 1e-3)
print("step-by-step kernel within 1e-4 of the PHILOX oracle:", np.mean(np.all(rel2 < 1e-4, axis=-1)), "trips", st2["wave_trips"])
