#!/usr/bin/env python3
"""What the mesh costs the mesh-bounded trackers: config 5's tracker scene (1024^3 cloud over the 320-triangle sphere, 4K)
rendered by the mesh-bounded integrator and by the plain one on the same grid, map and camera (FAST and FAST_LOCAL)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from volume_renderer_amd import MODE_FAST, MODE_FAST_LOCAL, Backend
from volume_renderer_amd import workloads as scenes

n, W, H = 1024, 3840, 2160
with Backend(0) as be:
    be.set_seed(scenes.SEED)
    be.generate_cloud(n, scenes.cloud_radius(n))
    be.set_density_map(*scenes.mesh_tracker_grid_map(n))
    be.set_mesh(scenes.icosphere_mesh())
    for integ, plain, spp, regime in (("meshDeltaTracking", "deltaTracking", 64, "textbook"), ("meshRatioTracking", "ratioTracking", 16, "shipped")):
        o = scenes.options_mesh_tracker(integ, n, W, H, spp, regime=regime)
        for mode, name in ((MODE_FAST, "fast"), (MODE_FAST_LOCAL, "local")):
            be.set_mode(mode)
            for who in (integ, plain):
                be.render(o, who)
                be.render(o, who)
                st = be.last_stats()
                print(f"{regime:8s} {name:5s} {who:18s} kernel {st['kernel_ms']:8.1f} ms  queries/sample {st['ray_queries'] / st['path_samples']:.2f}  "
                      f"lookups/sample {st['density_lookups'] / st['path_samples']:.1f} scatters/sample {st['scatter_events'] / st['path_samples']:.2f} "
                      f"trips {st['wave_trips']} phases {st['event_phases']}", flush=True)
