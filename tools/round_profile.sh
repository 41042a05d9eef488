#!/bin/bash
# GPU-box script: tests, the default bench, and the rocprofv3 passes whose summaries go to profiles/ (tools/collect_profiles.py).
# The profiler passes run the bench without its extra two-frames-in-flight leg (--pipeline off): every launch in the trace is then one
# of the one-frame-at-a-time launches the JSON line's value and roofline are measured on.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests_gpu.log 2>&1 || { tail -30 $O/tests_gpu.log; exit 1; }
tail -3 $O/tests_gpu.log
fi
timeout -k 10 600 python -u bench.py --steps 3 --warmup 1 > $O/bench_default.log 2>&1 || { tail -20 $O/bench_default.log; exit 1; }
grep '^{"metric"' $O/bench_default.log | tail -1 | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_kt $O/prof_fetch $O/prof_write $O/prof_sq $O/prof_sq2
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-baseline off --pipeline off > $O/prof_kt.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-baseline off --pipeline off > $O/prof_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-baseline off --pipeline off > $O/prof_write.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $O/prof_sq -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-baseline off --pipeline off > $O/prof_sq.log 2>&1 || exit 1
echo profiles done
