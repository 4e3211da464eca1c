#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
BALM_WINDOW_TRACE=1 timeout 300 python tools/bench_window.py 2> gpurun_out/r03k_trace.txt | tail -2
grep "add_scan" gpurun_out/r03k_trace.txt | tail -12
