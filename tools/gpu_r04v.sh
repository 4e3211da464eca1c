#!/bin/bash
# Round 4: K4 as one launch WITHOUT tickets (k_reduce_assemble) -- bitwise parity with the two-launch K4, the whole suite, timings at every size.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04v; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "one_launch_k4" > $OUT/pytest_k4.txt 2>&1 < /dev/null; echo "pytest k4 rc=$?"; tail -15 $OUT/pytest_k4.txt
for f in 1 0 1 0; do
  BALM_K4_FUSED=$f timeout 300 python bench.py --no-cpu --no-strong-ref --steps 60 2>/dev/null < /dev/null > $OUT/b.json
  echo "k4 fused=$f  $(grep -o '"ms_per_step": [0-9.]*' $OUT/b.json | head -1)  $(grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/b.json | cut -c1-150)" | tee -a $OUT/k4_ab.txt
done
for f in 1 0; do
  echo "== BALM_K4_FUSED=$f" | tee -a $OUT/k4_ab.txt
  BALM_K4_FUSED=$f timeout 300 python tools/bench_small.py 2>&1 < /dev/null | tail -4 | tee -a $OUT/k4_ab.txt
  BALM_K4_FUSED=$f timeout 300 python tools/bench_realshape.py 2>&1 < /dev/null | grep "sparse solve default" | cut -c1-200 | tee -a $OUT/k4_ab.txt
done
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
