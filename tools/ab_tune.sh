#!/bin/bash
# sweep of the round parameters (env tunables), same box, one process per point; "0 0" = built-in defaults
for w in "config2_textbook 64" "config3_textbook 256" "config4_delta_4k_256spp_cloud1024 32"; do
  read WL SPP <<< "$w"
  for cfg in "0 0" "96 32" "96 40" "192 32" "96 48" "0 0"; do
    set -- $cfg
    if [ $1 = 0 ]; then unset VR_TUNE_ROUND_MAX VR_TUNE_ROUND_EVENTS; else export VR_TUNE_ROUND_MAX=$1 VR_TUNE_ROUND_EVENTS=$2; fi
    timeout -k 10 120 python -u bench.py --workload $WL --spp $SPP --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$WL round_max $1 events $2', round(d['ms_per_step'],1), 'ms lanes', round(d['divergence']['walking_lanes_per_trip'],1), d['divergence']['wave_trips'], d['divergence']['event_phases'])" || exit 1
  done
done
