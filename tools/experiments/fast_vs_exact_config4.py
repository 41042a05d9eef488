import sys, numpy as np, torch
sys.path.insert(0, '.')
from volume_renderer_amd import MODE_EXACT, MODE_FAST, Backend
from volume_renderer_amd import workloads as scenes
n, W, H, spp = 1024, 3840, 2160, 64
o = scenes.options_cloud("deltaTracking", n, W, H, spp, regime="textbook")
fr = {}
with Backend(0) as be:
    be.set_seed(scenes.SEED); be.generate_cloud(n, scenes.cloud_radius(n))
    for m, name in ((MODE_FAST, "fast"), (MODE_EXACT, "exact")):
        be.set_mode(m); fr[name] = np.array(be.render(o, "deltaTracking"), dtype=np.float64)
        print(name, be.last_stats()["kernel_ms"], fr[name].mean(), flush=True)
d = (fr["fast"] - fr["exact"]).mean(axis=2).ravel()
print("fast - exact rel", d.mean() / fr["exact"].mean(), "z", d.mean() / (d.std(ddof=1) / np.sqrt(d.size)), "pixels equal within 1e-4:", np.mean(np.abs(fr["fast"] - fr["exact"]) <= 1e-4 * np.abs(fr["exact"]) + 1e-7))
