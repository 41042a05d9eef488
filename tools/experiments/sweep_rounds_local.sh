#!/bin/bash
# GPU-box script: round limits of the local-majorant kernel (W = workload, EVS = waiting lanes that end a round, RMS = trips per round)
W=${W:-config3_textbook}
for ev in ${EVS:-36 44 52}; do for rm in ${RMS:-64}; do
VR_TUNE_ROUND_EVENTS=$ev VR_TUNE_ROUND_MAX=$rm timeout -k 10 200 python bench.py --workload $W --mode local --steps 3 --warmup 1 --cpu-baseline off --pipeline off 2>&1 | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$W local events=$ev max=$rm', round(d['value'],1), round(d['roofline']['kernel_ms_per_launch'],2), round(d['divergence']['walking_lanes_per_trip'],1))" || exit 1
done; done
