#!/bin/bash
# Fourteenth GPU call of round 4: k_cov_factors one-pass (rows in registers across the block-wide sum, coalesced staged stores) against the two-pass kernel.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04n; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_cov.py -q -m gpu -x > $OUT/pytest_cov.txt 2>&1 < /dev/null; echo "pytest cov (one-pass) rc=$?"; tail -3 $OUT/pytest_cov.txt
BALM_COV_ONEPASS=0 timeout 900 python -m pytest tests/test_gpu_cov.py -q -m gpu -x > $OUT/pytest_cov2.txt 2>&1 < /dev/null; echo "pytest cov (two-pass) rc=$?"; tail -3 $OUT/pytest_cov2.txt
for m in 1 0 1 0; do
  echo "== BALM_COV_ONEPASS=$m" | tee -a $OUT/cov_ab.txt
  BALM_COV_ONEPASS=$m timeout 600 python tools/bench_cov.py 2>&1 < /dev/null | tail -6 | cut -c1-250 | tee -a $OUT/cov_ab.txt
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "store_paths" > $OUT/pytest_store.txt 2>&1 < /dev/null; echo "pytest store paths rc=$?"; tail -2 $OUT/pytest_store.txt
