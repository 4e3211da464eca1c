#!/bin/bash
# round 3: K2 with the clusters loaded two features ahead -- parity, then the bench's kernel table
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "evaluate or wide or damping_iter or subranges" 2>&1 | tail -4
for f in 0; do
  BALM_FUSE_TRIAL=$f timeout 600 python bench.py --no-cpu --no-accept --steps 30 --warmup 5 2>/dev/null | tee gpurun_out/r03h_bench_fuse$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('kernel_ms_per_step'))"
done
timeout 300 python tools/bench_realshape.py 2>&1 | tail -12
