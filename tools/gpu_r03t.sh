#!/bin/bash
# round 3: the solve's paths by window size after k_ldl_backsolve, and what the helpers of k_ldl_chain do
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_solve.py tests/test_gpu_parity.py tests/test_north_star.py tests/test_gpu_cov.py tests/test_gpu_multi.py -q -m gpu -x 2>&1 | tail -3
{ echo "# tools/bench_solve.py on MI355X, device ms per balm_solve_damped (HIP events); chain+backsolve = k_ldl_chain on [A ; rhs] + k_ldl_backsolve (default for 31..66 panels)"
  timeout 900 python tools/bench_solve.py 2>&1 | cut -c1-230; } | tee gpurun_out/r03t_solve_paths_by_window.txt
{ echo "# tools/chain_helpers.py (BALM_SOLVE_TRACE=1): the helper workgroups of k_ldl_chain during one factorisation"
  timeout 600 python tools/chain_helpers.py 2>&1 | tail -8; } | tee gpurun_out/r03t_chain_helpers.txt
