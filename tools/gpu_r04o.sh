#!/bin/bash
# Fifteenth GPU call of round 4: k_feature_factors (coalesced stores) with the clusters loaded two features ahead (BALM_FACTORS_DEPTH=2) -- A/B at config 2.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04o; mkdir -p $OUT
: > $OUT/depth_ab.txt
for rep in 1 2; do
  for d in 1 2; do
    BALM_FACTORS_DEPTH=$d timeout 300 python bench.py --no-cpu --no-strong-ref --steps 60 2>/dev/null < /dev/null > $OUT/b.json
    echo "rep $rep  depth=$d  $(grep -o '"ms_per_step": [0-9.]*' $OUT/b.json | head -1)  $(grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/b.json | cut -c1-150)" | tee -a $OUT/depth_ab.txt
  done
done
BALM_FACTORS_DEPTH=2 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $OUT/pytest_depth2.txt 2>&1 < /dev/null; echo "pytest (depth 2) rc=$?"; tail -2 $OUT/pytest_depth2.txt
