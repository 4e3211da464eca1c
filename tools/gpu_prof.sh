#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command (N=1, config 3)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd /tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu > $REPO/gpurun_out/prof/bench_stdout.txt 2>&1
cd $REPO
ls -R gpurun_out/prof | head -30
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
echo "== $f"; head -40 "$f"
