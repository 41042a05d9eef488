#!/bin/bash
# sweep of one env tunable over the mesh workloads, same box
for w in "config5_marcher_sphere_4k 16" "config5_marcher_nee_sphere_4k 16"; do
  read WL SPP <<< "$w"
  for v in 4 1 2 3 6 8 4; do
    VR_TUNE_BVH_LEAF=$v timeout -k 10 120 python -u bench.py --workload $WL --spp $SPP --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$WL leaf $v', round(d['ms_per_step'],2), 'ms', d['config']['frame_mean_rgb'][0])" || exit 1
  done
done
