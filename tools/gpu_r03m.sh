#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_window.py tests/test_window_golden.py tests/test_gpu_cov.py tests/test_gpu_voxel.py -q -m gpu -x 2>&1 | tail -4
timeout 300 python tools/bench_window.py 2>&1 | tail -1
timeout 300 python tools/count_window_launches.py 2>&1 | tail -1
timeout 300 python tools/bench_voxel.py --real 2>&1 | tail -4
BALM_WINDOW_TRACE=1 timeout 300 python tools/bench_window.py 2> gpurun_out/r03m_trace.txt | tail -1
grep "add_scan" gpurun_out/r03m_trace.txt | tail -4
