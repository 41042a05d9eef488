#!/bin/bash
# local-majorant mode: majorant cell size (VR_TUNE_MAJORANT_LEVEL l = cells of 8 << l voxels)
for l in ${LEVELS:-0 1 2 3 4}; do
  for w in "config3_textbook 64" "config4_delta_4k_256spp_cloud1024 16"; do set -- $w
    VR_TUNE_MAJORANT_LEVEL=$l timeout -k 10 200 python -u bench.py --workload $1 --spp $2 --mode local --steps 2 --warmup 1 --cpu-baseline off --pipeline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); dv=d['divergence']; print('level $l $1', round(d['value'],1), round(d['roofline']['kernel_ms_per_launch'],2), 'lookups/sample', round(d['lookups_per_sample'],2), 'trips', dv['wave_trips'])" || exit 1
  done
done
