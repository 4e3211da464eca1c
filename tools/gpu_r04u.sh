#!/bin/bash
# Round 4: k_reduce_all's tile sums with eight loads in flight (same order of additions); k_feature_factors grid sweep with the register accumulators.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04u; mkdir -p $OUT
for g in 0 384 640 1024 0; do
  if [ $g = 0 ]; then unset BALM_FACTORS_GRID; else export BALM_FACTORS_GRID=$g; fi
  timeout 300 python bench.py --no-cpu --no-strong-ref --steps 60 2>/dev/null < /dev/null > $OUT/b.json
  echo "factors grid=$g  $(grep -o '"ms_per_step": [0-9.]*' $OUT/b.json | head -1)  $(grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/b.json | cut -c1-150)" | tee -a $OUT/grid_ab.txt
done
unset BALM_FACTORS_GRID
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $OUT/pytest.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -2 $OUT/pytest.txt
