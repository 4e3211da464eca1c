#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/graph
echo "== graph tests"; timeout 900 python -m pytest tests/test_gpu_graph.py -m gpu -q -x 2>&1 | tail -12
echo "== small windows"; timeout 600 python tools/bench_small.py 2>&1 | tee gpurun_out/graph/bench_small.txt
