#!/bin/bash
# GPU-box script: the three PMC passes (HBM reads, HBM writes + L2 hits/misses, SQ instruction counters) for workloads other than
# the default one.  Entries "workload mode"; one launch per pass (--steps 1 --warmup 0).  tools/collect_pmc_workloads.py turns
# gpurun_out/pmc_wl/ into profiles/<round>_pmc_workloads.json and the entries of profiles/pmc_traffic.json that bench.py reads.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_wl
[ -z "$ENTRIES" ] && rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# ENTRIES="workload mode;workload mode" overrides the list (and keeps the files of earlier runs)
DEFAULT="config2_textbook fast;config3_delta_1080p_256spp_cloud512 fast;config3_textbook fast;config3_textbook local;config4_delta_4k_256spp_cloud1024 fast;config4_delta_4k_256spp_cloud1024 local;config5_tracker_ratio_sphere_4k fast;config5_tracker_delta_sphere_4k fast;config5_tracker_delta_sphere_4k local;config5_tracker_delta_shipped_sphere_4k fast;gridmarch_simple_1080p_cloud512 fast;config1_marcher_cube_512_1spp fast;config5_marcher_sphere_4k fast;config5_marcher_nee_sphere_4k fast;config5_marcher_cornell_4k fast;config5_marcher_nee_cornell_4k fast;perlin_marcher_sphere_1080p fast;small fast"
IFS=';' read -ra LIST <<< "${ENTRIES:-$DEFAULT}"
for e in "${LIST[@]}"; do
  set -- $e
  w=$1; m=$2; tag=${w}_${m}
  B="python3 $R/bench.py --workload $w --mode $m --steps 1 --warmup 0 --cpu-baseline off --pipeline off"
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$tag/fetch -- $B > $O/$tag.fetch.log 2>&1 || { tail -5 $O/$tag.fetch.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/$tag/write -- $B > $O/$tag.write.log 2>&1 || { tail -5 $O/$tag.write.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $O/$tag/sq -- $B > $O/$tag.sq.log 2>&1 || { tail -5 $O/$tag.sq.log; exit 1; }
  grep '^{"metric"' $O/$tag.sq.log | tail -1 > $O/${tag}_bench.json
  for p in fetch write sq; do f=$(find $O/$tag/$p -name "*counter_collection.csv" | head -1); cp "$f" $O/${tag}_$p.csv; done
  rm -rf $O/$tag
  echo "== $tag done"
done
