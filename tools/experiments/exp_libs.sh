#!/bin/bash
# Run config 2 (or WLS) on each experiment library given as argument and print the divergence counters it redefines
cp volume_renderer_amd/csrc/libvr_backend.so /tmp/base.so
for lib in /tmp/base.so "$@"; do
  cp $lib volume_renderer_amd/csrc/libvr_backend.so
  for w in ${WLS:-config2_ratio_1080p_64spp_cloud512}; do
    timeout -k 10 120 python -u bench.py --workload $w --steps 2 --warmup 1 --cpu-baseline off --pipeline off ${SPP:+--spp $SPP} 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); v=d['divergence']; print('$lib $w', 'kernel ms', round(d['roofline']['kernel_ms_per_launch'],2), 'trips', v['wave_trips'], 'lanes/trip', round(v['walking_lanes_per_trip'],2), 'phases', v['event_phases'])" || exit 1
  done
done
cp /tmp/base.so volume_renderer_amd/csrc/libvr_backend.so
