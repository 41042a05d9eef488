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
            b.set_seed(0xABC000 + 31 * k); fr.append(b.render(o, 3).astype(np.float64))
        res[mode] = np.stack(fr)
    a, c = res[MODE_FAST], res[MODE_FAST_LOCAL]
    den = np.sqrt((a.var(0, ddof=1) + c.var(0, ddof=1)) / K + (1e-5 * a.mean(0)) ** 2)
    z = (c.mean(0) - a.mean(0)) / np.maximum(den, 1e-30)
    ma, mc = a.mean(), c.mean()
    se = math.sqrt((a.mean(axis=(1, 2, 3)).var(ddof=1) + c.mean(axis=(1, 2, 3)).var(ddof=1)) / K)
    print(name, 'fast mean %.5f local mean %.5f rel %.5f (z of image mean %.1f)  pixel z: mean %.3f std %.3f <4 %.4f' % (ma, mc, (mc - ma) / ma, (mc - ma) / se, z.mean(), z.std(), np.mean(abs(z) < 4)), flush=True)
cloud = mg.cloud()
o = mg.options(3, "textbook", 1024)
cmp('cloud64 scatter', cloud, o)
cmp('cloud64 absorb-only', cloud, copy_options(o, sigma_a=1.0, sigma_s=0.0))
cmp('cloud64 scatter-only', cloud, copy_options(o, sigma_a=0.0, sigma_s=1.0))
homog = Grid(np.full((64, 64, 64), 0.5, np.float32))
cmp('homog scatter', homog, o)
cmp('homog scatter-only g=0', homog, copy_options(o, sigma_a=0.0, sigma_s=1.0, heyneyGreensteinG=0.0))
half = np.full((64, 64, 64), 0.5, np.float32); half[:, :, :32] = 0.05
cmp('two-halves scatter', Grid(half), o)
