#!/bin/bash
# GPU-box script: GPU tests, default bench, the config-5 tracker workloads
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG:-r2d}
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests_gpu.log 2>&1 || { tail -40 $O/tests_gpu.log; exit 1; }
tail -3 $O/tests_gpu.log
timeout -k 10 600 python -u bench.py --steps 3 --warmup 1 > $O/bench_default.log 2>&1 || { tail -20 $O/bench_default.log; exit 1; }
grep '^{"metric"' $O/bench_default.log | tail -1 | cut -c1-700
for w in "config5_tracker_ratio_sphere_4k 4" "config5_tracker_delta_sphere_4k 16" "config5_tracker_delta_shipped_sphere_4k 8"; do
  set -- $w
  timeout -k 10 300 python -u bench.py --workload $1 --spp $2 --steps 2 --warmup 1 --cpu-baseline off > $O/$1.log 2>&1 || { tail -20 $O/$1.log; exit 1; }
  grep '^{"metric"' $O/$1.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 spp$2', 'Msamples/s', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'lookups/sample', round(d['lookups_per_sample'],1), 'loads/lookup', round(d['roofline']['loads']['loads_per_lookup'],4), 'lanes', round(d['divergence']['walking_lanes_per_trip'],1), 'mean', [round(x,3) for x in d['config']['frame_mean_rgb']])"
done
