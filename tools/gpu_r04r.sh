#!/bin/bash
# Round 4: the NaN-robustness tests of the solve paths, then the whole suite once more.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04r; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_solve.py -q -m gpu -x -k "nan or non_finite" > $OUT/pytest_nan.txt 2>&1 < /dev/null; echo "pytest nan rc=$?"; tail -15 $OUT/pytest_nan.txt
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
