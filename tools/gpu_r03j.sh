#!/bin/bash
# round 3, VERDICT item 4: staged recut (device-side counts, one sync per stage, moved-point lists)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_window.py tests/test_window_golden.py tests/test_gpu_cov.py tests/test_gpu_voxel.py -q -m gpu -x 2>&1 | tail -6
for m in levels staged; do
  echo "== BALM_WINDOW_RECUT=$m"
  BALM_WINDOW_RECUT=$m timeout 300 python tools/bench_window.py 2>&1 | tail -2
  BALM_WINDOW_RECUT=$m timeout 300 python tools/count_window_launches.py 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp
for m in staged; do
  BALM_WINDOW_RECUT=$m timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r03j_win_$m -o w -- python $REPO/tools/count_window_launches.py > /dev/null 2>&1
  python $REPO/tools/rocpd_stats.py $REPO/gpurun_out/r03j_win_$m/w_results.db > $REPO/gpurun_out/r03j_win_${m}_kernel_stats.csv
  python - $REPO/gpurun_out/r03j_win_${m}_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
c=sum(int(r['Calls']) for r in rows); t=sum(float(r['TotalDurationNs']) for r in rows)
print('%d launches = %.1f per add_scan, device busy %.3f ms per add_scan' % (c, c/64, t/64e6))
PY
done
