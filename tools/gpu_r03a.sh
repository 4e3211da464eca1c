#!/bin/bash
# round 3, first GPU call: whole -m gpu suite (new comparators included), then the default bench line (writes the N=1 trace
# of configs[3] under gpurun_out/)
TAG=r03a
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 1200 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/$TAG/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/$TAG/pytest_gpu.txt
tail -40 gpurun_out/$TAG/pytest_gpu.txt
echo skip bench

