#!/bin/bash
# round 2, first GPU call: sync/latency microbenchmarks for the solve redesign + north-star acceptance tests + full GPU suite
mkdir -p gpurun_out/r02a
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
echo "== ubench_sync"; timeout 120 tools/bin/ubench_sync 2>&1 | tee gpurun_out/r02a/ubench_sync.txt
echo "== north star"; timeout 1500 python -m pytest tests/test_north_star.py -m gpu -q -x -s 2>&1 | tail -25 | tee gpurun_out/r02a/north_star.txt
echo "== pytest gpu (rest)"; timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_north_star.py 2>&1 | tail -8 | tee gpurun_out/r02a/pytest_gpu.txt
