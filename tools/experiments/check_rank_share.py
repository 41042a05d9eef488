import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, scenes
from volume_renderer_amd import Backend, MODE_FAST, INTEGRATORS
be=Backend(0); be.set_mode(MODE_FAST); be.set_seed(scenes.SEED); be.generate_cloud(128, scenes.cloud_radius(128))
o=scenes.options_cloud("ratioTracking",128,192,108,512)
full=be.render(o,"ratioTracking")
rows=len(range(3,108,8))
dev=torch.zeros((rows,192,3),dtype=torch.float32,device="cuda:0")
be.render_row_set_device(o,INTEGRATORS["ratioTracking"],3,8,rows,dev.data_ptr()); be.synchronize()
print("rank 3 of 8 at 512 spp equals the full frame's rows:", np.array_equal(full[3::8].view(np.uint32), dev.cpu().numpy().view(np.uint32)), full.mean())
# and the per-rank launch of the real N=8 weak case: 135 rows of 1080p at 512 spp
be.generate_cloud(512, scenes.cloud_radius(512))
o=scenes.options_cloud("ratioTracking",512,1920,1080,512)
dev=torch.zeros((135,1920,3),dtype=torch.float32,device="cuda:0")
t=time.perf_counter(); be.render_row_set_device(o,INTEGRATORS["ratioTracking"],3,8,135,dev.data_ptr()); be.synchronize(); dt=time.perf_counter()-t
print("one rank's share of the N=8 weak frame: %.1f ms, %.1f M samples/s per GPU"%(dt*1e3, 135*1920*512/dt/1e6), be.last_stats()["kernel_launches"])
