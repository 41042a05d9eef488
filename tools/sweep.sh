#!/bin/bash
# developer sweep of the step-loop round parameters and chunk at the headline workload (one process per point)
for cfg in "48 12 128" "48 8 128" "48 16 128" "32 12 128" "96 12 128" "48 12 64" "48 12 256" "48 12 128"; do
  set -- $cfg
  VR_TUNE_ROUND_MAX=$1 VR_TUNE_ROUND_EVENTS=$2 VR_TUNE_CHUNK=$3 timeout -k 10 120 python -u bench.py --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('round_max $1 events $2 chunk $3', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],1), 'ms lanes', round(d['divergence']['walking_lanes_per_trip'],1), d['divergence']['wave_trips'], d['divergence']['event_phases'])"
done
