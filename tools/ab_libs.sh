#!/bin/bash
# GPU-box script: A/B of libvr_backend.so builds.  args = alternative libraries (the tree's build is the base);
# WLS = "workload[:mode] ..." (default: config 2), REPS = repetitions (default 2)
cp volume_renderer_amd/csrc/libvr_backend.so /tmp/base.so
for rep in $(seq 1 ${REPS:-2}); do
for lib in /tmp/base.so "$@"; do
  cp $lib volume_renderer_amd/csrc/libvr_backend.so
  for wm in ${WLS:-config2_ratio_1080p_64spp_cloud512}; do
    w=${wm%%:*}; m=fast; [[ $wm == *:* ]] && m=${wm##*:}
    timeout -k 10 200 python -u bench.py --workload $w --mode $m --steps ${STEPS:-3} --warmup 1 --cpu-baseline off --pipeline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wm $lib', 'kernel ms', round(d['roofline']['kernel_ms_per_launch'],2), 'Msamples/s', round(d['value'],1), 'lanes/trip', round(d['divergence']['walking_lanes_per_trip'],1), d['config']['frame_mean_rgb'][0])" || { cp /tmp/base.so volume_renderer_amd/csrc/libvr_backend.so; exit 1; }
  done
done; done
cp /tmp/base.so volume_renderer_amd/csrc/libvr_backend.so
