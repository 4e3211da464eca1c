#!/bin/bash
# quick GPU iteration: parity tests + bench (no CPU leg) + kernel stats
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15
echo "== bench"; cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | grep "^{" | tee $REPO/gpurun_out/bench_prof.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"], d[\"kernel_ms_per_step\"], d[\"roofline\"][\"frac\"])"
cd $REPO; python tools/rocpd_stats.py gpurun_out/prof/bench_results.db | tee gpurun_out/kernel_stats.csv | head -24
