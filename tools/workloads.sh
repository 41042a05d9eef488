#!/bin/bash
# quick survey of the other BASELINE workloads at reduced spp (one process each)
for w in "config2_textbook 16" "config3_delta_1080p_256spp_cloud512 16" "config3_textbook 16" "config4_delta_4k_256spp_cloud1024 4" "config1_marcher_cube_512_1spp 1" "config5_marcher_sphere_4k 4" "config5_marcher_nee_sphere_4k 4"; do
  set -- $w
  timeout -k 10 300 python -u bench.py --workload $1 --spp $2 --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 spp$2', 'Msamples/s', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'kernel_ms', round(d['roofline']['kernel_ms_per_launch'],2), 'Glookups/s', round(d['glookups_per_s'],1), 'lookups/sample', round(d['lookups_per_sample'],1), 'lanes', round(d['divergence']['walking_lanes_per_trip'],1), 'trips', d['divergence']['wave_trips'], 'scat', d['divergence']['scatter_events'], 'mean', [round(x,3) for x in d['config']['frame_mean_rgb']])"
done
