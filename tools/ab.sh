#!/bin/bash
# A/B builds of libvr_backend.so on the same box: args = alternative libraries; env W="workload spp" (default: headline)
set -- "$@"
W=${W:-"config2_ratio_1080p_64spp_cloud512 64"}
read WL SPP <<< "$W"
cp volume_renderer_amd/csrc/libvr_backend.so /tmp/base.so
for rep in 1 2; do
for lib in /tmp/base.so "$@"; do
  cp $lib volume_renderer_amd/csrc/libvr_backend.so
  timeout -k 10 120 python -u bench.py --workload $WL --spp $SPP --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$WL $lib', round(d['ms_per_step'],1), round(d['lookups_per_sample']), round(d['divergence']['walking_lanes_per_trip'],1), d['divergence']['wave_trips'], d['config']['frame_mean_rgb'][0])"
done; done
cp /tmp/base.so volume_renderer_amd/csrc/libvr_backend.so
