#!/bin/bash
# Seventh GPU call of round 4: k_solve_small as the blocked, LDS-resident solve (chain wavefront + riders per diagonal tile) -- tests and timings.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04g; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_solve.py -q -m gpu -x -k "small" > $OUT/pytest_small.txt 2>&1 < /dev/null; echo "pytest small rc=$?"; tail -15 $OUT/pytest_small.txt
timeout 300 python tools/bench_solve.py 4 8 12 16 20 24 28 > $OUT/solve_small.txt 2>&1 < /dev/null; cut -c1-30,100-250 $OUT/solve_small.txt
timeout 300 python tools/bench_small.py > $OUT/small.txt 2>&1 < /dev/null; tail -5 $OUT/small.txt
