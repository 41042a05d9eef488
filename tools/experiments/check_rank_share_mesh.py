"""One rank's share of config 5 (mesh marcher, 4K) under an 8-way row interleave, against an eighth of the whole frame."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, scenes
from volume_renderer_amd import Backend, MODE_FAST, INTEGRATORS
be = Backend(0); be.set_mode(MODE_FAST); be.set_seed(scenes.SEED)
be.set_mesh(scenes.icosphere_mesh())
W, H, spp = 3840, 2160, 64
o = scenes.options_marcher(W, H, spp)
for name in ("homogeneousRayMarcherImproved", "homogeneousRayMarcherNEE"):
    integ = INTEGRATORS[name]
    full = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    part = torch.zeros((H // 8, W, 3), dtype=torch.float32, device="cuda:0")
    for rep in range(2):
        t = time.perf_counter(); be.render_row_set_device(o, integ, 0, 1, H, full.data_ptr()); be.synchronize(); t_full = time.perf_counter() - t
        t = time.perf_counter(); be.render_row_set_device(o, integ, 3, 8, H // 8, part.data_ptr()); be.synchronize(); t_part = time.perf_counter() - t
    same = bool((full[3::8] == part).all())
    print(f"{name}: full {t_full*1e3:.1f} ms, rank 3 of 8 {t_part*1e3:.1f} ms = {t_full/8/t_part*100:.0f} % of ideal, rows identical: {same}")
