#!/bin/bash
# round 3, item 5: the one-pass trial evaluation (k_moments_factors) -- parity, then A/B in the bench
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "one_pass or damping_iter or fixtures" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_multi.py tests/test_north_star.py -q -m gpu -x -k "not 8_shards" 2>&1 | tail -5
for f in 0 1; do
  echo "== BALM_FUSE_TRIAL=$f"
  BALM_FUSE_TRIAL=$f timeout 600 python bench.py --no-cpu --no-accept --steps 30 --warmup 5 2>/dev/null | tee gpurun_out/r03g_bench_fuse$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('kernel_ms_per_step'))"
done
