#!/usr/bin/env python3
"""Full-size cross-check of BASELINE config 2 (1920x1080, 512^3 cloud, ratio tracking, shipped step parameters): the oracle in the
reference's own scheme (libm, mt19937 per row) at K sub-samples per pixel over EVERY row, against the GPU's 64-spp frames
(FAST and EXACT).  bench.py's cpu_baseline leg makes the same comparison at K = 1; this is the longer look (K = argv[1], default 6:
about a minute on the GPU box's 16 cores).  Prints the image means, their relative difference and z = mean difference over its
standard error across pixels."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import oracle_lib as ol
from volume_renderer_amd import MODE_EXACT, MODE_FAST, Backend, INTEGRATORS
from volume_renderer_amd import workloads as scenes
from volume_renderer_amd.abi import Grid, copy_options

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n, W, H, spp = 512, 1920, 1080, 64
integ = "ratioTracking"
opts = scenes.options_cloud(integ, n, W, H, spp, regime="shipped")
frames = {}
with Backend(0) as be:
    be.set_seed(scenes.SEED)
    be.generate_cloud(n, scenes.cloud_radius(n))
    for mode, name in ((MODE_FAST, "fast"), (MODE_EXACT, "exact")):
        be.set_mode(mode)
        frames[name] = np.array(be.render(opts, integ), dtype=np.float64)
        print(name, "kernel ms", round(be.last_stats()["kernel_ms"], 1), "mean", frames[name].mean(), flush=True)

cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
L = ol.lib()
L.vro_set_math_mode(ol.MATH_LIBM)
vox = np.zeros((n, n, n), dtype=np.float32)
L.vro_make_cloud(n, scenes.cloud_radius(n), ol.fptr(vox), cores)
scene, _keep = ol.make_scene(density=Grid(vox))
oK = copy_options(opts, rayPerPixelCount=K)
cpu = np.zeros((H, W, 3), dtype=np.float64)


def one(j):
    cpu[j] = ol.render_rows(scene, oK, INTEGRATORS[integ], j, j + 1, rng_mode=ol.RNG_REFERENCE)[0][0]
    if j % 135 == 0:
        print("row", j, flush=True)


t0 = time.perf_counter()
with ThreadPoolExecutor(cores) as ex:
    list(ex.map(one, range(H)))
print(f"oracle: {K} sub-samples per pixel, {H} rows, {cores} threads, {time.perf_counter() - t0:.1f} s, mean {cpu.mean()}")
for name, f in frames.items():
    d = (cpu - f).mean(axis=2).ravel()
    se = d.std(ddof=1) / np.sqrt(d.size)
    print(f"{name:5s}: gpu mean {f.mean():.6f} oracle mean {cpu.mean():.6f} rel diff {(cpu.mean() - f.mean()) / f.mean():+.5f} "
          f"(standard error {se / f.mean():.5f}) z {d.mean() / se:+.2f}")
d = (frames["fast"] - frames["exact"]).mean(axis=2).ravel()
print(f"fast - exact: rel diff {d.mean() / frames['exact'].mean():+.6f}, z {d.mean() / (d.std(ddof=1) / np.sqrt(d.size)):+.2f}")
