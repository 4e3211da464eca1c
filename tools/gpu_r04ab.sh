#!/bin/bash
# Round 4: k_reduce_all's tile sums with four lanes per element on small tile sets -- parity, timings over feature counts, the whole suite.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04ab; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "four_lanes" > $OUT/pytest_quads.txt 2>&1 < /dev/null; echo "pytest quads rc=$?"; tail -4 $OUT/pytest_quads.txt
for m in 1 0; do
  echo "== BALM_REDUCE_QUADS=$m" | tee -a $OUT/quads_ab.txt
  BALM_REDUCE_QUADS=$m timeout 300 python tools/bench_w20.py 20 2>&1 < /dev/null | tee -a $OUT/quads_ab.txt
  BALM_REDUCE_QUADS=$m timeout 300 python tools/bench_w20.py 64 2>&1 < /dev/null | tee -a $OUT/quads_ab.txt
done
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
