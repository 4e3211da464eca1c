#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
BALM_BENCH_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace -d $REPO/gpurun_out/r03x_dist -o d -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu --no-accept > /dev/null 2>&1
python - $REPO/gpurun_out/r03x_dist/d_results.db <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
extra = [c for c in cols if c in ("stream_id", "queue_id", "stream", "queue", "tid")]
rows = cur.execute("select %s, start, end %s from kernels order by start" % (name, "".join(", " + e for e in extra))).fetchall()
# last LM iteration: find the last k_hessian_syrk and print 40 kernels around it
idx = [i for i, r in enumerate(rows) if "k_hessian_syrk" in r[0]]
i0 = idx[-2] - 12
t0 = rows[i0][1]
prev_end = t0
for r in rows[i0:i0 + 60]:
    nm = r[0].replace("(anonymous namespace)::", "").split("(")[0][:60]
    print("%9.1f us  +%7.1f gap  %8.1f us  %s %s" % ((r[1] - t0) / 1e3, (r[1] - prev_end) / 1e3, (r[2] - r[1]) / 1e3, nm, r[3:]))
    prev_end = max(prev_end, r[2])
PY
rm -rf $REPO/gpurun_out/r03x_dist
