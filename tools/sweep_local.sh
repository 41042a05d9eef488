#!/bin/bash
# sweep of the round limits (events that end a round / trips per round); MODE=local|fast, EVS, RMS lists
for ev in ${EVS:-48 56}; do for rm in ${RMS:-64 128}; do
  for w in "config3_textbook 64" "config4_delta_4k_256spp_cloud1024 16" ${EXTRA:+"$EXTRA"}; do set -- $w
    VR_TUNE_ROUND_EVENTS=$ev VR_TUNE_ROUND_MAX=$rm timeout -k 10 200 python -u bench.py --workload $1 --spp $2 --mode ${MODE:-local} --steps 2 --warmup 1 --cpu-baseline off --pipeline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); dv=d['divergence']; print('${MODE:-local} events $ev max $rm $1', round(d['value'],1), round(d['roofline']['kernel_ms_per_launch'],2), 'trips', dv['wave_trips'], 'phases', dv['event_phases'], 'lanes', round(dv['walking_lanes_per_trip'],1))"
  done
done; done
