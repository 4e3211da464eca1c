#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_solve.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python tools/bench_solve.py 64 100 200 224 240 256 280 300 350 400 480 500 600 700 800 2>&1 | tee gpurun_out/r03s_solve.txt
