"""Developer experiment: do two frames on two backends (two streams) overlap the first one's tail?"""
import sys, time
sys.path.insert(0, '.')
import torch
from volume_renderer_amd import Backend, MODE_FAST, INTEGRATORS
from volume_renderer_amd import workloads as wl
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, W, H = 512, 1920, 1080
bes = [Backend(0), Backend(0)]
bufs = [torch.zeros((H, W, 3), dtype=torch.float32, device='cuda:0') for _ in bes]
o = wl.options_cloud("ratioTracking", n, W, H, spp)
for b in bes:
    b.set_mode(MODE_FAST); b.set_seed(wl.SEED); b.generate_cloud(n, wl.cloud_radius(n))
def frame(b, buf):
    b.render_row_set_device(o, 2, 0, 1, H, buf.data_ptr())
for b, buf in zip(bes, bufs):
    frame(b, buf); b.synchronize()
K = 6
t = time.perf_counter()
for k in range(K):
    frame(bes[0], bufs[0]); bes[0].synchronize()
serial = (time.perf_counter() - t) / K
t = time.perf_counter()
for k in range(K):
    frame(bes[k % 2], bufs[k % 2])
    if k > 0: bes[(k - 1) % 2].synchronize()
bes[(K - 1) % 2].synchronize()
piped = (time.perf_counter() - t) / K
print(f"spp {spp}: serial {serial*1e3:.1f} ms/frame, two in flight {piped*1e3:.1f} ms/frame, gain {(serial/piped-1)*100:.1f} %")
assert torch.equal(bufs[0], bufs[1])
