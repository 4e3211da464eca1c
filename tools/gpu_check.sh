#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/check
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 120 tools/bin/ubench_f64 2>&1 | tee gpurun_out/check/ubench_f64.txt
