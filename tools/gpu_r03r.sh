#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_window.py tests/test_window_golden.py tests/test_gpu_cov.py tests/test_gpu_voxel.py tests/test_gpu_multi.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -4
timeout 300 python tools/bench_window.py 2>&1 | tail -1
timeout 300 python tools/bench_voxel.py --real 2>&1 | tail -3
