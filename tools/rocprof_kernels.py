#!/usr/bin/env python3
"""Per-dispatch kernel times out of a rocprofv3 results database (<dir>/**/*_results.db).  rocprofv3 on this stack often
segfaults at process exit AFTER writing the database and BEFORE writing its CSV summaries (profiles/README.md); the database is
complete, so read that.     python tools/rocprof_kernels.py <dir-or-db> [name-filter]
Prints: kernel, duration (us), gap to the previous dispatch (us), grid x workgroup, dynamic LDS, scratch per lane; then
per-kernel averages."""
import glob
import os
import re
import sqlite3
import sys


def main():
    if len(sys.argv) < 2:
        print(__doc__); return 2
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*results.db"), recursive=True))
    if not dbs:
        print("no *results.db under %s" % path); return 1
    for dbp in dbs:
        cur = sqlite3.connect(dbp).cursor()
        try:
            rows = list(cur.execute("select name, start, end, grid_x, workgroup_x, lds_size, scratch_size from kernels order by start"))
        except sqlite3.Error as e:
            print("%s: %s" % (dbp, e)); continue
        print("# %s: %d dispatches" % (dbp, len(rows)))
        prev, agg = None, {}
        for name, s, e, gx, wx, lds, scr in rows:
            short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
            short = re.sub(r"^void ", "", short)
            if flt in short and not short.startswith("__amd_rocclr"):
                gap = "%8.1f" % ((s - prev) / 1000.0) if prev is not None and s - prev < 5e6 else "       -"
                print("%-34s %10.1f %s   %d x %d, %d B, %d B" % (short[:34], (e - s) / 1000.0, gap, gx // max(wx, 1), wx, lds, scr))
                a = agg.setdefault(short, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1000.0
            prev = e
        print("# averages")
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print("%-34s %6d x %10.1f us = %10.1f us" % (k[:34], n, t / n, t))
    return 0


if __name__ == "__main__":
    sys.exit(main())
