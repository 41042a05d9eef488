#!/bin/bash
# local-majorant mode on configs 3-textbook and 4 with the tree's build
for w in config3_textbook config4_delta_4k_256spp_cloud1024; do
  timeout -k 10 200 python -u bench.py --workload $w --mode local --steps 3 --warmup 1 --cpu-baseline off --pipeline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; v=d['divergence']; print('$w', 'kernel ms', round(r['kernel_ms_per_launch'],2), 'M/s', round(d['value'],1), 'lookups/sample', round(d['lookups_per_sample'],2), 'trips', v['wave_trips'], 'lanes', round(v['walking_lanes_per_trip'],1), d['config']['frame_mean_rgb'])" || exit 1
done
