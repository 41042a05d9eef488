#!/bin/bash
# one SQ counter pass over a short run of the headline workload; prints the tracking kernel's row
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_a $O/pmc_b
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_a -- python3 $R/bench.py --spp ${SPP:-8} --workload ${WL:-config2_ratio_1080p_64spp_cloud512} --steps 1 --warmup 0 --cpu-baseline off > $O/pmc_a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_b -- python3 $R/bench.py --spp ${SPP:-8} --workload ${WL:-config2_ratio_1080p_64spp_cloud512} --steps 1 --warmup 0 --cpu-baseline off > $O/pmc_b.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
for d in ("$O/pmc_a","$O/pmc_b"):
    agg=collections.defaultdict(float)
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "${KERNEL:-vr_track_kernel}" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in sorted(agg.items()): print(k, "%.4g"%v)
PY
