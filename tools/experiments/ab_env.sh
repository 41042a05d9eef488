#!/bin/bash
# GPU-box script: one environment knob off / on over a list of workloads.  usage: KNOB=VR_TUNE_X WLS="workload[:mode] ..." ab_env.sh
for wm in ${WLS:-config2_ratio_1080p_64spp_cloud512}; do for v in "" 1; do
  w=${wm%%:*}; m=fast; [[ $wm == *:* ]] && m=${wm##*:}
  env $KNOB=$v timeout -k 10 200 python -u bench.py --workload $w --mode $m --steps ${STEPS:-3} --warmup 1 --cpu-baseline off --pipeline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wm $KNOB=$v', 'kernel ms', round(d['roofline']['kernel_ms_per_launch'],2), 'Msamples/s', round(d['value'],1), d['config']['frame_mean_rgb'][0])" || exit 1
done; done
