#!/bin/bash
# per-launch timeline of the launch-path factorisation at one window size (kernel trace)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-480}
mkdir -p $R/gpurun_out/soltrace; cd /tmp
BALM_SOLVE=launches timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/soltrace -o t -- python $R/tools/bench_solve_one.py $W > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob
f = glob.glob('gpurun_out/soltrace/**/t_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last solve: find the last k_rank_diag or k_build_A
idx = [i for i, r in enumerate(rows) if 'k_build_A' in r['Kernel_Name']]
last = rows[idx[-1]:]
t0 = int(last[0]['Start_Timestamp'])
prev_end = t0
out = []
for r in last:
    n = r['Kernel_Name'].split('(')[0].replace('balm::', '').replace('void ', '')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append((n, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
import collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for n, s, d, g in out:
    agg[n][0] += 1; agg[n][1] += d; agg[n][2] += g
print("kernel, launches, total us, total gap-before us")
for n, v in agg.items(): print("%-28s %4d %9.1f %9.1f" % (n, v[0], v[1], v[2]))
print("span us", out[-1][1] + out[-1][2])
pt = [o for o in out if 'panel_trail' in o[0]]
print("k_ldl_panel_trail durations (us) every 6th:", [round(o[2], 1) for o in pt[::6]])
tr = [o for o in out if o[0].startswith('k_ldl_trail')]
print("k_ldl_trail (A) durations every 6th:", [round(o[2], 1) for o in tr[::6]], "gaps:", [round(o[3], 1) for o in tr[::6]])
PY
