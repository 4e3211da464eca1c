#!/bin/bash
# round 3, VERDICT item 4: balm_window_add_scan with one recut pass for all levels and a lookup-first cut_voxel
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_window.py tests/test_window_golden.py tests/test_gpu_cov.py -q -m gpu -x 2>&1 | tail -6
for m in levels merged; do
  echo "== BALM_WINDOW_RECUT=$m"
  BALM_WINDOW_RECUT=$m timeout 300 python tools/bench_window.py 2>&1 | tail -2
  BALM_WINDOW_RECUT=$m timeout 300 python tools/count_window_launches.py 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp
for m in levels merged; do
  BALM_WINDOW_RECUT=$m timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r03i_win_$m -o w -- python $REPO/tools/count_window_launches.py > /dev/null 2>&1
  f=$(ls $REPO/gpurun_out/r03i_win_$m/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -z "$f" ] && f=$(ls $REPO/gpurun_out/r03i_win_$m/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $m: $f"
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['Calls']) for r in rows); ns=sum(float(r['TotalDurationNs']) for r in rows)
print("kernel launches %d = %.1f per add_scan (64 scans); device time %.3f ms per add_scan" % (tot, tot/64.0, ns/64e6))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:8]: print("  %-70s calls %6s  total %.3f ms" % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6))
PY
done
