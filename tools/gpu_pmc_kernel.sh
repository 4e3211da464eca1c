#!/bin/bash
# PMC counters of one workload, per kernel, one rocprofv3 pass per counter group (groups separated by '/'):
#   tools/gpu_pmc_kernel.sh <tag> "SQ_WAVES SQ_INSTS_VALU / SQ_INSTS_LDS" <command...>
REPO=$(pwd); TAG=$1; shift; GROUPS_=$1; shift; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
cd /tmp
i=0
IFS='/' read -ra GS <<< "$GROUPS_"
for g in "${GS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/pmc/g$i -o p -- "$@" > $OUT/pass$i.log 2>&1
  tail -2 $OUT/pass$i.log | cut -c1-200
done
cd $REPO
python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.csv 2>&1; rm -rf $OUT/pmc
head -14 $OUT/pmc_summary.csv | cut -c1-300
