#!/bin/bash
# Seventeenth GPU call of round 4: k_ldl_chain with the wait for the panel's stores behind B3 (wave 8 publishes) -- solve tests, timings, the chain's timeline.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04q; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_solve.py -q -m gpu -x > $OUT/pytest_solve.txt 2>&1 < /dev/null; echo "pytest solve rc=$?"; tail -3 $OUT/pytest_solve.txt
timeout 600 python tools/bench_solve.py 40 64 100 177 200 256 300 500 800 > $OUT/solve.txt 2>&1 < /dev/null; cut -c1-30,100-250 $OUT/solve.txt
BALM_SOLVE_TRACE=1 timeout 300 python tools/chain_check.py 200 > $OUT/chain_trace_n1200.txt 2>&1 < /dev/null; sed -n 1,16p $OUT/chain_trace_n1200.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_north_star.py tests/test_gpu_cov.py -q -m gpu -x > $OUT/pytest_more.txt 2>&1 < /dev/null; echo "pytest more rc=$?"; tail -3 $OUT/pytest_more.txt
