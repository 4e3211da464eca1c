#!/bin/bash
# Fourth GPU call of round 4: K2 (k_feature_factors) with the lane's pose in registers (3 workgroups per CU instead of 2) and/or streaming stores for Gt,
# each twice, alternating; then the parity suites that cover the factors.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04d; mkdir -p $OUT
: > $OUT/factors_ab.txt
for rep in 1 2; do
  for cfg in "PREG=1 NT=0" "PREG=0 NT=0" "PREG=1 NT=1" "PREG=0 NT=1"; do
    set -- $cfg; preg=${1#PREG=}; nt=${2#NT=}
    BALM_FACTORS_PREG=$preg BALM_GT_NT=$nt timeout 300 python bench.py --no-cpu --no-strong-ref --steps 60 2>/dev/null < /dev/null > $OUT/b.json
    echo "rep $rep  pose-in-registers=$preg  streaming-stores=$nt  $(grep -o '"ms_per_step": [0-9.]*' $OUT/b.json | head -1)  $(grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/b.json | cut -c1-150)" >> $OUT/factors_ab.txt
  done
done
cat $OUT/factors_ab.txt
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.txt
