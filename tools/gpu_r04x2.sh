#!/bin/bash
# Round 4: plan_syrk's small-window rule (shortest waves that fill one round) against the old >= 64-k-step rule (BALM_SYRK_SMALL=0), over feature counts.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04x; mkdir -p $OUT
for m in 1 0; do
  echo "== BALM_SYRK_SMALL=$m" | tee -a $OUT/small_ab.txt
  BALM_SYRK_SMALL=$m timeout 300 python tools/bench_w20.py 20 2>&1 < /dev/null | tee -a $OUT/small_ab.txt
  BALM_SYRK_SMALL=$m timeout 300 python tools/bench_w20.py 64 2>&1 < /dev/null | tee -a $OUT/small_ab.txt
  BALM_SYRK_SMALL=$m timeout 300 python tools/bench_small.py 2>&1 < /dev/null | tail -4 | tee -a $OUT/small_ab.txt
  BALM_SYRK_SMALL=$m timeout 600 python tools/bench_window.py 2>&1 < /dev/null | tail -1 | cut -c1-260 | tee -a $OUT/small_ab.txt
done
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu2.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu2.txt
