#!/bin/bash
# full GPU suite + round artifacts.  Usage: gpu_round.sh r02f
TAG=${1:-rXX}
mkdir -p gpurun_out/$TAG
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/$TAG/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/gpu_profile_round.sh $TAG
