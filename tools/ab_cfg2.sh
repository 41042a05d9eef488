#!/bin/bash
# config 2 (and its textbook variant) on the tree's build, twice: kernel ms, rate, frame mean
for rep in 1 2; do
for w in ${WLS:-config2_ratio_1080p_64spp_cloud512 config2_textbook}; do
  timeout -k 10 120 python -u bench.py --workload $w --steps 3 --warmup 1 --cpu-baseline off --pipeline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', 'ms', round(d['ms_per_step'],2), 'kernel', round(d['roofline']['kernel_ms_per_launch'],2), 'Msamples/s', round(d['value'],1), d['config']['frame_mean_rgb'])" || exit 1
done; done
