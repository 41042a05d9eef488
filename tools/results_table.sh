#!/bin/bash
# the BASELINE.md results table: every config at its full size on one GPU, CPU baseline beside it
mkdir -p gpurun_out/results
for w in "config1_marcher_cube_512_1spp 5" "config2_ratio_1080p_64spp_cloud512 20" "config2_textbook 10" "config3_delta_1080p_256spp_cloud512 20" "config3_textbook 10" "config4_delta_4k_256spp_cloud1024 10" "config5_marcher_sphere_4k 10" "config5_marcher_nee_sphere_4k 10" "gridmarch_simple_1080p_cloud512 10"; do
  set -- $w
  echo "== $1"
  timeout -k 10 900 python -u bench.py --workload $1 --steps 2 --warmup 1 --cpu-seconds $2 2>/dev/null | tail -1 > gpurun_out/results/$1.json || echo "FAILED $1"
  python3 -c "import json; d=json.loads(open('gpurun_out/results/$1.json').read()); c=d.get('cpu_baseline') or {}; print(round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],1), 'ms', round(d['glookups_per_s'],1), 'Glookups/s frac', round(d['roofline']['frac'],3), 'cpu', c.get('value'), c.get('cores'))"
done
