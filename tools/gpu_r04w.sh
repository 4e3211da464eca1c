#!/bin/bash
# Round 4, closing evidence beside r04z: the kernels of a 20-pose window's LM iteration (rocprofv3 kernel trace of 60 iterations), the other BASELINE sizes and the
# extreme shapes on the final tree.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04w; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "20 20" "20 150"; do
  set -- $cfg
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_$1_$2 -o t -- python $REPO/tools/small_lm_trace.py $1 $2 > /dev/null 2>&1 )
  db=$(find $OUT/trace_$1_$2 -name "*_results.db" | head -1)
  echo "== W=$1 F=$2: kernels of three LM runs of 20 iterations (60 iterations)" | tee -a $OUT/small_lm_kernels.txt
  timeout 120 python tools/rocpd_stats.py $db 2>/dev/null | head -14 | tee -a $OUT/small_lm_kernels.txt
  rm -rf $OUT/trace_$1_$2
done
timeout 900 bash tools/gpu_configs.sh > $OUT/other_configs.txt 2>&1 < /dev/null; cat $OUT/other_configs.txt | cut -c1-260
timeout 900 python tools/stress_shapes.py > $OUT/stress_shapes.txt 2>&1 < /dev/null; cat $OUT/stress_shapes.txt | cut -c1-200
