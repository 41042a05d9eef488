# GPU-box script: bench.py's cpu_baseline.cross_check (oracle rows against the frame the bench rendered) over the workloads of WLS
mkdir -p gpurun_out/cc
for wm in ${WLS:-config3_delta_1080p_256spp_cloud512:fast config3_textbook:fast config3_textbook:local config2_textbook:fast config5_marcher_sphere_4k:fast config5_marcher_nee_cornell_4k:fast config5_tracker_delta_sphere_4k:fast config5_tracker_delta_sphere_4k:local config5_tracker_ratio_sphere_4k:fast config4_delta_4k_256spp_cloud1024:fast config4_delta_4k_256spp_cloud1024:local}; do
  w=${wm%%:*}; m=${wm##*:}
  timeout -k 10 400 python -u bench.py --workload $w --mode $m --steps 1 --warmup 1 --pipeline off --cpu-baseline on > gpurun_out/cc/${w}_$m.json 2> gpurun_out/cc/${w}_$m.err || { echo "FAILED $wm"; tail -3 gpurun_out/cc/${w}_$m.err; exit 1; }
  python3 -c "
import json,sys; d=json.loads(open('gpurun_out/cc/${w}_$m.json').read().strip().splitlines()[-1]); c=d['cpu_baseline']; print('$wm', round(d['value'],1), c['cross_check'], c['sample'][-40:])"
done
