#!/bin/bash
for ev in ${EVS:-12 16 24 32}; do for rm in ${RMS:-0}; do
  for w in "config5_tracker_ratio_sphere_4k 8" "config5_tracker_delta_shipped_sphere_4k 16" "config5_tracker_delta_sphere_4k 32"; do set -- $w
    VR_TUNE_ROUND_EVENTS=$ev VR_TUNE_ROUND_MAX=$rm timeout -k 10 200 python -u bench.py --workload $1 --spp $2 --steps 2 --warmup 1 --cpu-baseline off --pipeline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); dv=d['divergence']; print('events $ev max $rm $1', round(d['value'],1), round(d['roofline']['kernel_ms_per_launch'],2), 'trips', dv['wave_trips'], 'phases', dv['event_phases'], 'lanes', round(dv['walking_lanes_per_trip'],1))"
  done
done; done
