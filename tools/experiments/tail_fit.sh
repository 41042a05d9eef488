#!/bin/bash
# launch time of the headline kernel against spp: the intercept is the tail of the longest paths
for spp in 4 8 16 32 64; do
  timeout -k 10 120 python -u bench.py --spp $spp --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print($spp, round(d['roofline']['kernel_ms_per_launch'],2), 'ms  lanes', round(d['divergence']['walking_lanes_per_trip'],1))" || exit 1
done
