#!/bin/bash
# sweep of one VR_TUNE_* knob on the same box, one process per point (0 = built-in default)
# usage: VAR=VR_TUNE_CHUNK W="workload spp" tools/experiments/ab_tune.sh 0 64 128 256
VAR=${VAR:-VR_TUNE_ROUND_MAX}
W=${W:-"config2_ratio_1080p_64spp_cloud512 64"}
read WL SPP <<< "$W"
for v in "$@"; do
  if [ $v = 0 ]; then unset $VAR; else export $VAR=$v; fi
  timeout -k 10 120 python -u bench.py --workload $WL --spp $SPP --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$WL $VAR $v', round(d['ms_per_step'],1), 'ms lanes', round(d['divergence']['walking_lanes_per_trip'],1), d['divergence']['event_phases'])" || exit 1
done
