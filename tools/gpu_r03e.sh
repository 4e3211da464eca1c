#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 900 python -m pytest tests/test_gpu_window.py tests/test_gpu_cov.py -q -m gpu -x -k "consistency" -s 2>&1 | tail -15
