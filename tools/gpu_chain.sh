#!/bin/bash
# k_ldl_chain bring-up: correctness vs LAPACK + times; every python process under its own timeout (bounded waits in the kernel)
mkdir -p gpurun_out/chain
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 300 python tools/chain_check.py 64 177 200 256 > gpurun_out/chain/check.txt 2>&1; echo "rc=$?" >> gpurun_out/chain/check.txt
cat gpurun_out/chain/check.txt
BALM_SOLVE_TRACE=1 timeout 120 python tools/chain_check.py 200 > gpurun_out/chain/trace200.txt 2>&1; echo "rc=$?" >> gpurun_out/chain/trace200.txt
tail -60 gpurun_out/chain/trace200.txt
