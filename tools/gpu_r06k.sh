#!/bin/bash
REPO=$(pwd); TAG=${TAG:-r06k}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
timeout 1800 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.txt
BALM_BENCH_LOOPBACK=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu 2>$OUT/bench_loopback2.err | grep "^{" > $OUT/bench_loopback2.json; echo "loopback bench rc=$?"; cut -c1-1200 $OUT/bench_loopback2.json; tail -3 $OUT/bench_loopback2.err
