#!/bin/bash
# solve iteration: fused-vs-launches parity, then timing at the bench size and at other window sizes
mkdir -p gpurun_out/solve
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
echo "== solve tests"; timeout 600 python -m pytest tests/test_gpu_solve.py -m gpu -q -x 2>&1 | tail -15
echo "== timing"; timeout 600 python tools/bench_solve.py 64 80 100 128 177 200 256 300 350 400 2>&1 | tee gpurun_out/solve/bench_solve.txt
