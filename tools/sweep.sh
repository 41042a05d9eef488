#!/bin/bash
# developer sweep of the step-loop tuning parameters (one process per point)
for tb in 1 2 3 4; do for rm in 12 24; do for ev in 6 12; do
  VR_TUNE_TRIP_BLOCKS=$tb VR_TUNE_ROUND_EVENTS=$ev VR_TUNE_ROUND_MAX=$rm timeout -k 10 120 python -u bench.py --spp 16 --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('blocks $tb round_max $rm events $ev', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],1), 'ms lanes', round(d['divergence']['walking_lanes_per_trip'],1), d['divergence']['wave_trips'], d['divergence']['event_phases'])"
done; done; done
