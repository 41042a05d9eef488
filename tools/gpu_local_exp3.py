import math, os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden'); sys.path.insert(0, '.')
import numpy as np
import make_stat_golden as mg, scenes
from volume_renderer_amd import Backend, MODE_FAST, MODE_FAST_LOCAL, Grid, copy_options
b = Backend(0)
def cmp(name, grid, o, K=16):
    b.set_density(grid)
    res = {}
    for mode in (MODE_FAST, MODE_FAST_LOCAL):
        b.set_mode(mode); fr = []
        for k in range(K):
            b.set_seed((0xABC000 if mode == MODE_FAST else 0x123000) + 31 * k); fr.append(b.render(o, 3).astype(np.float64))
        res[mode] = np.stack(fr)
        st = b.last_stats()
    a, c = res[MODE_FAST], res[MODE_FAST_LOCAL]
    ma, mc = a.mean(), c.mean()
    se = math.sqrt((a.mean(axis=(1, 2, 3)).var(ddof=1) + c.mean(axis=(1, 2, 3)).var(ddof=1)) / K)
    print(name, 'fast mean %.5f local mean %.5f rel %.5f (z of image mean %.1f) scatters/path %.3f' % (ma, mc, (mc - ma) / ma, (mc - ma) / max(se, 1e-30), st['scatter_events'] / st['path_samples']), flush=True)
o = mg.options(3, "textbook", 1024)
cmp('homog 64^3 0.5', Grid(np.full((64, 64, 64), 0.5, np.float32)), o)
cmp('homog 8^3 vox8 4.0 (one brick)', Grid(np.full((8, 8, 8), 4.0, np.float32), voxel_size=8.0), o)
cmp('homog 16^3 vox4 2.0', Grid(np.full((16, 16, 16), 2.0, np.float32), voxel_size=4.0), o)
cmp('homog 64^3 maxDepth 1', Grid(np.full((64, 64, 64), 0.5, np.float32)), copy_options(o, maxDepth=1))
cmp('homog 64^3 maxDepth 2', Grid(np.full((64, 64, 64), 0.5, np.float32)), copy_options(o, maxDepth=2))
cmp('homog 64^3 maxDepth 255', Grid(np.full((64, 64, 64), 0.5, np.float32)), copy_options(o, maxDepth=255))
cmp('homog 64^3 g=0', Grid(np.full((64, 64, 64), 0.5, np.float32)), copy_options(o, heyneyGreensteinG=0.0))
