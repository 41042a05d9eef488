#!/bin/bash
# GPU-box script: round limits of the GEOM kernel on config 3 (and config 5's delta shipped leg)
W=${W:-config3_delta_1080p_256spp_cloud512}
for ev in ${EVS:-8 16 24 32 44}; do for rm in ${RMS:-96}; do
VR_TUNE_ROUND_EVENTS=$ev VR_TUNE_ROUND_MAX=$rm timeout -k 10 200 python bench.py --workload $W --steps 2 --warmup 1 --cpu-baseline off --pipeline off 2>&1 | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$W events=$ev max=$rm', round(d['value'],1), round(d['ms_per_step'],1), round(d['divergence']['walking_lanes_per_trip'],1), d['divergence']['wave_trips'], d['divergence']['event_phases'])" || exit 1
done; done
