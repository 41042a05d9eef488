#!/bin/bash
# A/B of libvr_backend.so builds on the mesh workloads: args = alternative libraries (the tree's build is the base)
cp volume_renderer_amd/csrc/libvr_backend.so /tmp/base.so
for rep in 1 2; do
for lib in /tmp/base.so "$@"; do
  cp $lib volume_renderer_amd/csrc/libvr_backend.so
  for w in ${WLS:-config5_marcher_sphere_4k config5_marcher_nee_sphere_4k config1_marcher_cube_512_1spp perlin_marcher_sphere_1080p config5_tracker_delta_shipped_sphere_4k}; do
    timeout -k 10 120 python -u bench.py --workload $w --steps 3 --warmup 1 --cpu-baseline off ${SPP:+--spp $SPP} 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $lib', 'ms', round(d['ms_per_step'],2), 'kernel', round(d['roofline']['kernel_ms_per_launch'],2), 'Msamples/s', round(d['value'],1), d['config']['frame_mean_rgb'][0])"
  done
done; done
cp /tmp/base.so volume_renderer_amd/csrc/libvr_backend.so
