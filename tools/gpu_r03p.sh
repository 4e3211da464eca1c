#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py tests/test_gpu_multi.py tests/test_gpu_solve.py tests/test_gpu_cov.py tests/test_gpu_sparse_syrk.py -q -m gpu -x 2>&1 | tail -4
timeout 300 python tools/bench_realshape.py 2>&1 | grep "shipped\|default" | tee gpurun_out/r03p_realshape.txt
timeout 300 python tools/bench_small.py 2>&1 | tail -8 | tee gpurun_out/r03p_small.txt
timeout 600 python bench.py --no-cpu --no-accept --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('kernel_ms_per_step'))"
