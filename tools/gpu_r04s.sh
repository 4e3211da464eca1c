#!/bin/bash
# Round 4: k_feature_factors with the lane's pose and accumulators in registers (BALM_FACTORS_REGS=1) on top of the coalesced stores -- A/B at config 2, parity.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04s; mkdir -p $OUT
: > $OUT/regs_ab.txt
for rep in 1 2; do
  for r in 0 1; do
    BALM_FACTORS_REGS=$r timeout 300 python bench.py --no-cpu --no-strong-ref --steps 60 2>/dev/null < /dev/null > $OUT/b.json
    echo "rep $rep  regs=$r  $(grep -o '"ms_per_step": [0-9.]*' $OUT/b.json | head -1)  $(grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/b.json | cut -c1-150)" | tee -a $OUT/regs_ab.txt
  done
done
BALM_FACTORS_REGS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $OUT/pytest_regs.txt 2>&1 < /dev/null; echo "pytest (regs) rc=$?"; tail -2 $OUT/pytest_regs.txt
