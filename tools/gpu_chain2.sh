#!/bin/bash
# k_ldl_chain: where it wins (selection table) + its tests
mkdir -p gpurun_out/chain
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 600 python tools/chain_check.py 20 24 32 40 48 64 100 128 144 177 200 256 300 320 350 400 480 500 700 > gpurun_out/chain/select.txt 2>&1; echo "rc=$?" >> gpurun_out/chain/select.txt
cat gpurun_out/chain/select.txt
timeout 900 python -m pytest tests/test_gpu_solve.py -q -m gpu -k "chain" 2>&1 | tail -5
