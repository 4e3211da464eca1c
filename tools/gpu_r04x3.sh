#!/bin/bash
# Round 4: the block-sparse plan's items fill one round of wave slots on small problems (runs of < 8 chunks) -- shipped window, sliding window, suite.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04x; mkdir -p $OUT
for m in 1 0 1 0; do
  echo "== BALM_SYRK_SMALL=$m" | tee -a $OUT/sparse_ab.txt
  BALM_SYRK_SMALL=$m timeout 300 python tools/bench_realshape.py 2>&1 < /dev/null | grep "solve default" | cut -c1-200 | tee -a $OUT/sparse_ab.txt
  BALM_SYRK_SMALL=$m timeout 600 python tools/bench_window.py 2>&1 < /dev/null | tail -1 | cut -c1-260 | tee -a $OUT/sparse_ab.txt
done
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu3.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu3.txt
