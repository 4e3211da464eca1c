#!/bin/bash
# first GPU contact: microbench, parity tests, small + full bench
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== ubench"; timeout 120 tools/bin/ubench_f64 2>&1 | tee gpurun_out/ubench.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.txt
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q --maxfail=8 -x 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.txt
echo "== bench small"; timeout 300 python bench.py --win 200 --features 5000 --steps 5 --warmup 2 --no-cpu 2>&1 | tail -3 | tee gpurun_out/bench_small.txt
echo "== bench full"; timeout 600 python bench.py --steps 10 --warmup 2 2>&1 | tail -3 | tee gpurun_out/bench_full.txt
