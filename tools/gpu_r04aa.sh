#!/bin/bash
# Round 4: rocprofv3 kernel tables of three more workloads -- a 64-pose window's LM iterations, the shipped window's, the sliding-window map's calls -- to see what bounds each.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04aa; mkdir -p $OUT
export TMPDIR=/tmp
run() {   # name, command...
  name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$name -o t -- "$@" > /dev/null 2>&1 )
  db=$(find $OUT/trace_$name -name "*_results.db" | head -1)
  echo "== $name" | tee -a $OUT/kernel_tables.txt
  timeout 120 python $REPO/tools/rocpd_stats.py $db 2>/dev/null | head -16 | cut -c1-150 | tee -a $OUT/kernel_tables.txt
  rm -rf $OUT/trace_$name
}
run w64_f5000 python $REPO/tools/small_lm_trace.py 64 5000
run w64_f1000 python $REPO/tools/small_lm_trace.py 64 1000
run shipped_window python $REPO/tools/bench_realshape.py
run sliding_window python $REPO/tools/bench_window.py --scans 40
