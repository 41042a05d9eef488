#!/bin/bash
# GPU-box script: rocprofv3 kernel-trace summaries of the other workloads at full size (copied to profiles/ afterwards)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_wl
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in config3_delta_1080p_256spp_cloud512 config3_textbook config2_textbook config4_delta_4k_256spp_cloud1024 config5_marcher_sphere_4k config5_marcher_nee_sphere_4k gridmarch_simple_1080p_cloud512; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --cpu-baseline off > $O/$w.log 2>&1 || { tail -5 $O/$w.log; exit 1; }
  f=$(find $O/$w -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/${w}_kernel_stats.csv
  grep '^{"metric"' $O/$w.log | tail -1 > $O/${w}_bench.json
  rm -rf $O/$w
  echo "== $w"; head -2 $O/${w}_kernel_stats.csv | tail -1 | cut -c1-160
done
