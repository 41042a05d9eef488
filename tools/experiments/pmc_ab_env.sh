#!/bin/bash
# GPU-box script: the FETCH_SIZE / SQ passes of one workload under two settings of an environment knob.
# usage: pmc_ab_env.sh KNOB "workload mode"   (runs KNOB= and KNOB=1)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_ab; rm -rf $O; mkdir -p $O
knob=$1; set -- $2; w=$1; m=$2
cd /tmp && export TMPDIR=/tmp
for v in "" 1; do
  export $knob=$v
  B="python3 $R/bench.py --workload $w --mode $m --steps 1 --warmup 0 --cpu-baseline off --pipeline off"
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$v -- $B > $O/f$v.log 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/w$v -- $B > $O/w$v.log 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $O/s$v -- $B > $O/s$v.log 2>&1 || exit 1
  for p in f w s; do
    f=$(find $O/$p$v -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$knob=$v" <<'P'
import csv, sys, collections
c = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if "vr_" in r["Kernel_Name"] and ("track" in r["Kernel_Name"] or "dda" in r["Kernel_Name"] or "march" in r["Kernel_Name"]):
        c[r["Counter_Name"]] += float(r["Counter_Value"])
print(sys.argv[2], {k: f"{v:.4g}" for k, v in c.items()})
P
  done
done
