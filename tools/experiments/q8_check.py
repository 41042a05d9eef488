#!/usr/bin/env python3
"""FAST_LOCAL frames with the byte-quantised companion grid against the same frames from the float grid alone (VR_TUNE_NO_Q8=1):
bit-identical by construction; prints the kernel times of both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from volume_renderer_amd import MODE_FAST_LOCAL, Backend
from volume_renderer_amd import workloads as scenes

for n, w, h, spp in ((64, 96, 80, 16), (128, 160, 120, 16), (512, 480, 270, 16)):
    with Backend(0) as be:
        be.set_seed(scenes.SEED)
        be.generate_cloud(n, scenes.cloud_radius(n))
        be.set_mode(MODE_FAST_LOCAL)
        o = scenes.options_cloud("deltaTracking", n, w, h, spp, regime="textbook")
        frames = {}
        for off in ("1", ""):
            os.environ["VR_TUNE_NO_Q8"] = off
            be.render(o, "deltaTracking")
            frames[off] = be.render(o, "deltaTracking").copy()
            st = be.last_stats()
            print(n, "no_q8" if off else "q8", "kernel ms", round(st["kernel_ms"], 3), "lookups", st["density_lookups"], "mean", float(frames[off].mean()))
        same = np.array_equal(frames["1"].view(np.uint32), frames[""].view(np.uint32))
        print(n, "bit-identical:", same)
        if not same:
            sys.exit(1)
