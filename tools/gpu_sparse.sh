#!/bin/bash
mkdir -p gpurun_out/sparse
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
echo "== sparse tests"; timeout 900 python -m pytest tests/test_gpu_sparse_syrk.py -m gpu -q -x 2>&1 | tail -15
echo "== real shape"; timeout 600 python tools/bench_realshape.py 2>&1 | tee gpurun_out/sparse/realshape.txt
echo "== realworld + parity regression"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
