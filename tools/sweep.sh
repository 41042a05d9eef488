#!/bin/bash
# developer sweep of the step-loop round parameters (one process per point)
for ev in 4 8 12 20; do for rm in 16 32 64; do
  VR_TUNE_ROUND_EVENTS=$ev VR_TUNE_ROUND_MAX=$rm timeout -k 10 120 python -u bench.py --spp 4 --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$ev $rm', round(d['ms_per_step'],1), round(d['divergence']['walking_lanes_per_trip'],1), d['divergence']['wave_trips'], d['divergence']['event_phases'])"
done; done
