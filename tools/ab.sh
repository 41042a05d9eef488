#!/bin/bash
# A/B builds of libvr_backend.so on the same box: args = alternative libraries
cp volume_renderer_amd/csrc/libvr_backend.so /tmp/base.so
for rep in 1 2; do
for lib in /tmp/base.so "$@"; do
  cp $lib volume_renderer_amd/csrc/libvr_backend.so
  timeout -k 10 120 python -u bench.py --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],1), round(d['lookups_per_sample']), round(d['divergence']['walking_lanes_per_trip'],1), d['divergence']['wave_trips'], d['config']['frame_mean_rgb'][0])"
done; done
cp /tmp/base.so volume_renderer_amd/csrc/libvr_backend.so
