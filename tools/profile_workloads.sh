#!/bin/bash
# GPU-box script: rocprofv3 kernel-trace summaries of the other workloads at full size (copied to profiles/ afterwards).
# Entries: "workload mode [spp]" (spp only where the full count would take minutes: stated in the JSON line's config.spp)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_wl
[ -z "$ONLY" ] && rm -rf $O; mkdir -p $O     # ONLY="workload mode;workload mode": just those entries, keeping the others' files
cd /tmp && export TMPDIR=/tmp
for e in "config3_delta_1080p_256spp_cloud512 fast" "config3_textbook fast" "config3_textbook local" "config5_tracker_delta_sphere_4k local" "config2_textbook fast" \
         "config4_delta_4k_256spp_cloud1024 fast" "config4_delta_4k_256spp_cloud1024 local" "config1_marcher_cube_512_1spp fast" \
         "config5_marcher_sphere_4k fast" "config5_marcher_nee_sphere_4k fast" "config5_marcher_cornell_4k fast" "config5_marcher_nee_cornell_4k fast" "config5_tracker_ratio_sphere_4k fast" \
         "config5_tracker_delta_sphere_4k fast" "config5_tracker_delta_shipped_sphere_4k fast" "gridmarch_simple_1080p_cloud512 fast" \
         "perlin_marcher_sphere_1080p fast"; do
  if [ -n "$ONLY" ] && [[ ";$ONLY;" != *";$e;"* ]]; then continue; fi
  set -- $e
  w=$1; m=$2; tag=${w}_${m}
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 $R/bench.py --workload $w --mode $m --steps 2 --warmup 1 --cpu-seconds 8 --pipeline off ${3:+--spp $3} > $O/$tag.log 2>&1 || { tail -5 $O/$tag.log; exit 1; }
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/${tag}_kernel_stats.csv
  grep '^{"metric"' $O/$tag.log | tail -1 > $O/${tag}_bench.json
  rm -rf $O/$tag
  echo "== $tag"; head -2 $O/${tag}_kernel_stats.csv | tail -1 | cut -c1-160
done
