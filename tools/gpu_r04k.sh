#!/bin/bash
# Eleventh GPU call of round 4: k_solve_small with H requested up front (coalesced, under the ranking) and scattered into the tiles from registers;
# tools/ubench_f64 with more write-only forms (is 4.09 TB/s the part's write rate or the grid-stride form's?); the whole suite.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04k; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_solve.py -q -m gpu -x -k "small" > $OUT/pytest_small.txt 2>&1 < /dev/null; echo "pytest small rc=$?"; tail -3 $OUT/pytest_small.txt
timeout 300 python tools/small_trace.py 8 16 20 24 > $OUT/small_trace.txt 2>&1 < /dev/null; cat $OUT/small_trace.txt | cut -c1-400
timeout 300 python tools/bench_solve.py 4 8 12 16 20 24 > $OUT/solve_small.txt 2>&1 < /dev/null; cut -c1-30,100-250 $OUT/solve_small.txt
timeout 300 python tools/bench_small.py > $OUT/small.txt 2>&1 < /dev/null; tail -5 $OUT/small.txt
timeout 200 tools/bin/ubench_f64 > $OUT/ubench_f64.txt 2>&1 < /dev/null; grep -i "copy\|read-only\|write-only" $OUT/ubench_f64.txt
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
