#!/bin/bash
# round 3, VERDICT item 4: final numbers for balm_window_add_scan -- call times, launches per call, kernel table
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_window.py tests/test_window_golden.py -q -m gpu -x 2>&1 | tail -3
{
for m in levels staged; do
  echo "== BALM_WINDOW_RECUT=$m (levels = the round-2 recut, one pass per octree level; staged = the default)"
  BALM_WINDOW_RECUT=$m timeout 300 python tools/bench_window.py 2>&1 | tail -1
  BALM_WINDOW_RECUT=$m timeout 300 python tools/count_window_launches.py 2>&1 | tail -1
done
} | tee gpurun_out/r03n_window.txt
cd /tmp && export TMPDIR=/tmp
for m in levels staged; do
  BALM_WINDOW_RECUT=$m timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r03n_win_$m -o w -- python $REPO/tools/count_window_launches.py > /dev/null 2>&1
  python $REPO/tools/rocpd_stats.py $REPO/gpurun_out/r03n_win_$m/w_results.db > $REPO/gpurun_out/r03n_window_${m}_kernel_stats.csv
  rm -rf $REPO/gpurun_out/r03n_win_$m
  python - $REPO/gpurun_out/r03n_window_${m}_kernel_stats.csv $m <<'PY' | tee -a $REPO/gpurun_out/r03n_window.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
c=sum(int(r['Calls']) for r in rows); t=sum(float(r['TotalDurationNs']) for r in rows)
print('%s (rocprofv3 --kernel-trace, 64 add_scan calls into a growing 64-scan window): %d launches = %.1f per add_scan, kernel time %.3f ms per add_scan' % (sys.argv[2], c, c/64, t/64e6))
PY
done
BALM_WINDOW_TRACE=1 timeout 300 python $REPO/tools/bench_window.py 2> $REPO/gpurun_out/r03n_trace.txt > /dev/null
grep "add_scan" $REPO/gpurun_out/r03n_trace.txt | tail -5 | tee -a $REPO/gpurun_out/r03n_window.txt
