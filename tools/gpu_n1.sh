#!/bin/bash
mkdir -p gpurun_out/n1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
echo "== build tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_north_star.py tests/test_gpu_cov.py -m gpu -q -x -k "build or virtual or consistency or nees or device_built" 2>&1 | tail -6
echo "== N1 bench"; timeout 600 python tools/bench_cluster_build.py 2>&1 | tee gpurun_out/n1/bench_cluster_build.txt
timeout 600 python tools/bench_cluster_build.py --pts 40 --features 3000 2>&1 | tee -a gpurun_out/n1/bench_cluster_build.txt
