"""Developer experiment: VR_MODE_FAST_LOCAL against FAST and the 4096-spp reference fixture (64^3 cloud, delta, textbook)."""
import math, os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden'); sys.path.insert(0, '.')
import numpy as np
import make_stat_golden as mg, stat_common as sc
from volume_renderer_amd import Backend, MODE_FAST, MODE_FAST_LOCAL, Grid, default_options
fx = np.load('tests/golden/stat_fixtures.npz')
b = Backend(0); b.set_density(mg.cloud())
o = mg.options(3, "textbook")
res = {}
for mode in (MODE_FAST, MODE_FAST_LOCAL):
    b.set_mode(mode)
    fr = []
    for k in range(32):
        b.set_seed(0xABC000 + 31 * k); fr.append(b.render(o, 3).astype(np.float64))
    res[mode] = np.stack(fr)
    print(mode, sc.compare_frames(res[mode], fx['delta_textbook_mean'], fx['delta_textbook_stderr']), b.last_stats()['density_lookups'] / (64 * 64 * 4096), flush=True)
a, c = res[MODE_FAST], res[MODE_FAST_LOCAL]
z = (a.mean(0) - c.mean(0)) / np.maximum(np.sqrt((a.var(0, ddof=1) + c.var(0, ddof=1)) / 32), 1e-30)
print('local vs fast: frac|z|<4 %.4f <3 %.4f mean z %.3f std %.3f  image mean rel %.5f' % (np.mean(abs(z) < 4), np.mean(abs(z) < 3), z.mean(), z.std(), (c.mean() - a.mean()) / a.mean()))
# slabs
def slab_options(spp, **kw):
    base = dict(width=48, height=48, widthReference=4800, heightReference=4800, widthStartOffset=2376, heightStartOffset=2376,
                rayPerPixelCount=spp, maxDepth=8, sigma_a=0.25, sigma_s=0.0, stepSizeMin=1e-4, stepSizeMax=1e30,
                stepSizeMultiplier=1.0, cameraPosition=(0, 0, 40.0), fov=30.0, backgroundColor=(1.0, 1.0, 1.0), mediumColor=(0.0, 0.0, 0.0))
    base.update(kw); return default_options(**base)
n = 16
vox = np.full((n, n, n), 0.2, np.float32); vox[:, :, 5:12] = 0.9
for name, v, sa, tau in (("homog", np.full((n, n, n), 0.4, np.float32), 0.25, 0.25 * 0.4 * 16), ("layered", vox, 0.3, 0.3 * (0.9 * 7 + 0.2 * 9))):
    b.set_density(Grid(v)); b.set_mode(MODE_FAST_LOCAL); b.set_seed(7)
    img = b.render(slab_options(1024, sigma_a=sa), 3)
    got = float(img[..., 0].astype(np.float64).mean()); want = math.exp(-tau)
    print(name, got, want, (got - want) / math.sqrt(want * (1 - want) / (48 * 48 * 1024)))
