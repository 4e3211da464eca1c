#!/bin/bash
# solve iteration: path parity (fused vs launches, lookahead on/off), then timing at the bench size and at other window sizes
mkdir -p gpurun_out/solve
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
echo "== solve tests"; timeout 600 python -m pytest tests/test_gpu_solve.py -m gpu -q -x 2>&1 | tail -15
echo "== timing"; timeout 600 python tools/bench_solve.py ${@:-64 128 200 256 300 350 400 480 600 800 1024} 2>&1 | tee gpurun_out/solve/bench_solve.txt
