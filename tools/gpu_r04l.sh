#!/bin/bash
# Twelfth GPU call of round 4: k_feature_factors with its Gt columns leaving as fully coalesced 16-byte stores through an LDS staging block per wavefront
# (BALM_FACTORS_STAGE=1) and / or contiguous feature ranges per workgroup (BALM_FACTORS_BLOCKED=1): A/B at config 2, parity under both.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04l; mkdir -p $OUT
: > $OUT/factors_ab.txt
for rep in 1 2; do
  for cfg in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfg
    BALM_FACTORS_STAGE=$1 BALM_FACTORS_BLOCKED=$2 timeout 300 python bench.py --no-cpu --no-strong-ref --steps 60 2>/dev/null < /dev/null > $OUT/b.json
    echo "rep $rep  staged=$1  blocked=$2  $(grep -o '"ms_per_step": [0-9.]*' $OUT/b.json | head -1)  $(grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/b.json | cut -c1-150)" >> $OUT/factors_ab.txt
  done
done
cat $OUT/factors_ab.txt
for g in 256 1024; do
  BALM_FACTORS_STAGE=1 BALM_FACTORS_GRID=$g timeout 300 python bench.py --no-cpu --no-strong-ref --steps 60 2>/dev/null < /dev/null > $OUT/b.json
  echo "staged=1 grid=$g  $(grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/b.json | cut -c1-150)" | tee -a $OUT/factors_ab.txt
done
BALM_FACTORS_STAGE=1 BALM_FACTORS_BLOCKED=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_north_star.py -q -m gpu -x > $OUT/pytest_stage.txt 2>&1 < /dev/null; echo "pytest (staged, blocked) rc=$?"; tail -3 $OUT/pytest_stage.txt
BALM_FACTORS_STAGE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_cov.py -q -m gpu -x > $OUT/pytest_stage2.txt 2>&1 < /dev/null; echo "pytest (staged) rc=$?"; tail -3 $OUT/pytest_stage2.txt
