#!/bin/bash
# Round 4: k_feature_factors with the lane's pose and accumulators in registers as the default for the left form -- parity (the two homes, the whole suite), bench.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04t; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "accumulator_homes or store_paths" > $OUT/pytest_k2.txt 2>&1 < /dev/null; echo "pytest k2 rc=$?"; tail -3 $OUT/pytest_k2.txt
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
for r in 1 0 1 0; do
  BALM_FACTORS_REGS=$r timeout 300 python bench.py --no-cpu --no-strong-ref --steps 60 2>/dev/null < /dev/null > $OUT/b.json
  echo "regs=$r  $(grep -o '"ms_per_step": [0-9.]*' $OUT/b.json | head -1)  $(grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/b.json | cut -c1-150)" | tee -a $OUT/regs_ab.txt
done
timeout 300 python tools/bench_realshape.py > $OUT/realshape.txt 2>&1 < /dev/null; grep "sparse solve default" $OUT/realshape.txt | cut -c1-220
timeout 300 python tools/bench_small.py > $OUT/small.txt 2>&1 < /dev/null; tail -4 $OUT/small.txt
