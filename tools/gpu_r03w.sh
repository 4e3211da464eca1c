#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
echo "== plain"
BALM_SOLVE_DEBUG=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-accept 2>gpurun_out/w_plain.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'])"; sort gpurun_out/w_plain.err | uniq -c | grep "balm_hip: solve" | head -3
echo "== forced dist, no torchrun"
BALM_BENCH_FORCE_DIST=1 BALM_SOLVE_DEBUG=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-accept 2>gpurun_out/w_dist.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step'])"; sort gpurun_out/w_dist.err | uniq -c | grep "balm_hip: solve" | head -3; grep -i "error\|fail" gpurun_out/w_dist.err | head -3
