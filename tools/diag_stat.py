import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, oracle_lib as ol, scenes
from volume_renderer_amd import *
L=ol.lib()
n=64
vox=np.zeros((n,n,n),dtype=np.float32); L.vro_make_cloud(n, scenes.cloud_radius(n), ol.fptr(vox), 8)
g=Grid(vox); sc,_=ol.make_scene(density=g)
W=H=40
be=Backend(0); be.set_density(g); be.set_seed(scenes.SEED)
for spp in (4,16,64,256):
    o=scenes.options_cloud(2,n,W,H,spp)
    L.vro_set_math_mode(1)
    want=ol.render(sc,o,2,rng_mode=0,seed=scenes.SEED)
    L.vro_set_math_mode(0)
    for mode,name in ((MODE_EXACT,'exact'),(MODE_EXACT_SKIP,'exact_skip'),(MODE_FAST,'fast')):
        be.set_mode(mode); got=be.render(o,2)
        rel=np.abs(got-want)/np.maximum(np.abs(want),1e-3)
        print(spp,name,'bitexact',np.array_equal(got.view(np.uint32),want.view(np.uint32)),'frac<1e-4',np.mean(np.all(rel<1e-4,axis=-1)),'mean',got.mean(),want.mean())
