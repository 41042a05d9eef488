#!/bin/bash
# developer sweep of the step-loop round parameters over the tracking workloads (one process per point)
for w in "config2_ratio_1080p_64spp_cloud512 32" "config3_delta_1080p_256spp_cloud512 16" "config3_textbook 16" "config4_delta_4k_256spp_cloud1024 4"; do
set -- $w
for ev in 8 12 20 32; do
  VR_TUNE_ROUND_EVENTS=$ev timeout -k 10 120 python -u bench.py --workload $1 --spp $2 --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 events $ev', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],1), 'ms lanes', round(d['divergence']['walking_lanes_per_trip'],1), d['divergence']['wave_trips'], d['divergence']['event_phases'])"
done; done
