#!/bin/bash
# Experiment: how much of a launch is its drain (trips after the work counters ran dry)?  tools/ab/drain.so is the
# library built with -DVR_DRAIN_EXP, whose divergence counters see only those trips.
cp volume_renderer_amd/csrc/libvr_backend.so /tmp/base.so
for lib in /tmp/base.so tools/ab/drain.so; do
  cp $lib volume_renderer_amd/csrc/libvr_backend.so
  for spp in 64 8; do
    timeout -k 10 120 python -u bench.py --steps 3 --warmup 1 --cpu-baseline off --pipeline off --spp $spp 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); v=d['divergence']; print('$lib spp $spp', 'kernel ms', round(d['roofline']['kernel_ms_per_launch'],2), 'trips', v['wave_trips'], 'lanes/trip', round(v['walking_lanes_per_trip'],2), 'phases', v['event_phases'])" || exit 1
  done
done
cp /tmp/base.so volume_renderer_amd/csrc/libvr_backend.so
