#!/bin/bash
# sweep of the round's trip limit per regime, same box, one process per point (0 = built-in default)
sweep() {
  read WL SPP <<< "$1"; shift
  for v in "$@"; do
    if [ $v = 0 ]; then unset VR_TUNE_ROUND_MAX; else export VR_TUNE_ROUND_MAX=$v; fi
    timeout -k 10 120 python -u bench.py --workload $WL --spp $SPP --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$WL round_max $v', round(d['ms_per_step'],1), 'ms lanes', round(d['divergence']['walking_lanes_per_trip'],1), d['divergence']['event_phases'])" || exit 1
  done
}
sweep "config2_ratio_1080p_64spp_cloud512 64" 0 24 32 64 96 0
sweep "config3_delta_1080p_256spp_cloud512 64" 0 16 24 32 96 0
