#!/bin/bash
# N1 (cluster build) after the tail prefetch + term-lane mapping: parity of every path, then the two bench shapes
mkdir -p gpurun_out/r03b
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "build_clusters" 2>&1 | tail -15
{
echo "# tools/bench_cluster_build.py on MI355X: 6-point runs (F=20000) and the launch default of 40-point runs (F=3000), W=200, 24 M points"
timeout 300 python tools/bench_cluster_build.py --features 20000 --pts 6
timeout 300 python tools/bench_cluster_build.py --features 3000 --pts 40
echo "# A/B: lane-per-run mapping forced on 40-point runs, term-lane mapping forced on 6-point runs"
BALM_BUILD_TERMS=0 timeout 300 python tools/bench_cluster_build.py --features 3000 --pts 40
BALM_BUILD_TERMS=1 timeout 300 python tools/bench_cluster_build.py --features 20000 --pts 6
echo "# 100-point and 1000-point runs"
timeout 300 python tools/bench_cluster_build.py --features 1200 --pts 100
timeout 300 python tools/bench_cluster_build.py --features 120 --pts 1000
} > gpurun_out/r03b/cluster_build.txt 2>&1
cat gpurun_out/r03b/cluster_build.txt
