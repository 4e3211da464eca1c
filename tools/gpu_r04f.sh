#!/bin/bash
# Sixth GPU call of round 4: k_solve_small (the whole damped solve of a window of <= 32 poses as one launch) for the first time -- its tests, the solve and
# LM-iteration timings against the launch path; the overlapped evaluation's three orders side by side (where does H differ, what does the cut alone cost).
REPO=$(pwd); OUT=$REPO/gpurun_out/r04f; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_solve.py -q -m gpu -x -k "small" > $OUT/pytest_small.txt 2>&1 < /dev/null; echo "pytest small rc=$?"; tail -15 $OUT/pytest_small.txt
timeout 300 python tools/bench_solve.py 4 8 12 16 20 24 28 32 > $OUT/solve_small.txt 2>&1 < /dev/null; cut -c1-30,100-250 $OUT/solve_small.txt
timeout 300 python tools/bench_small.py > $OUT/small.txt 2>&1 < /dev/null; tail -5 $OUT/small.txt
BALM_SOLVE=launches timeout 300 python tools/bench_small.py > $OUT/small_launches.txt 2>&1 < /dev/null; sed 's/^/solve=launches /' $OUT/small_launches.txt | head -2
timeout 300 python tools/exp_overlap.py 200 10000 > $OUT/exp_overlap.txt 2>&1 < /dev/null; tail -8 $OUT/exp_overlap.txt | cut -c1-260
for ovl in 2 0 1; do
  BALM_OVERLAP=$ovl timeout 300 python bench.py --no-cpu --no-strong-ref --steps 60 2>$OUT/b.err < /dev/null > $OUT/b_$ovl.json
  echo "overlap=$ovl  $(grep -o '"ms_per_step": [0-9.]*' $OUT/b_$ovl.json | head -1)  $(grep -o '"kernel_ms_per_step": {[^}]*}' $OUT/b_$ovl.json)" | tee -a $OUT/overlap_ab.txt
done
BALM_OVERLAP=0 timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_parity.py::test_overlapped_evaluation_is_the_serial_one > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.txt
