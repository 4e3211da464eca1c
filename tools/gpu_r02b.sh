#!/bin/bash
# round 2, second GPU call: in-library multi-device tests, full suite, bench (N=1 with the CPU leg, and the N>1 code path on one GPU)
mkdir -p gpurun_out/r02b
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
echo "== multi"; timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/r02b/multi.txt
echo "== pytest gpu (all)"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r02b/pytest_gpu.txt
echo "== bench N=1"; timeout 900 python bench.py 2>gpurun_out/r02b/bench.err | grep "^{" > gpurun_out/r02b/bench.json; cut -c1-1500 gpurun_out/r02b/bench.json
echo "== bench dist path on one GPU"; BALM_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu 2>gpurun_out/r02b/bench_dist.err | grep "^{" > gpurun_out/r02b/bench_dist.json; cut -c1-900 gpurun_out/r02b/bench_dist.json; tail -5 gpurun_out/r02b/bench_dist.err
