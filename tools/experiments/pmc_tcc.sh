#!/bin/bash
# TCC (L2) counter passes over a short run; KERNEL / WL / SPP as in pmc_sq.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_c $O/pmc_d
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_REQ_sum --output-format csv -d $O/pmc_c -- python3 $R/bench.py --spp ${SPP:-8} --workload ${WL:-config2_ratio_1080p_64spp_cloud512} --steps 1 --warmup 0 --cpu-baseline off > $O/pmc_c.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/pmc_d -- python3 $R/bench.py --spp ${SPP:-8} --workload ${WL:-config2_ratio_1080p_64spp_cloud512} --steps 1 --warmup 0 --cpu-baseline off > $O/pmc_d.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
for d in ("$O/pmc_c","$O/pmc_d"):
    agg=collections.defaultdict(float)
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "${KERNEL:-vr_track_kernel}" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in sorted(agg.items()): print(k, "%.4g"%v)
PY
