#!/usr/bin/env python3
"""Does the link warm up?  The shipped window's balm_associate upload (215 MB) call after call: back to back, with the LM run in
between (as the end-to-end leg does), and with 50 ms of idle time in between.  Prints the BALM_T_UPLOAD span of every call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from balm_amd import capi, realworld as rw
d = np.load(rw.SHIPPED_WINDOW_NPZ)
xyz = np.ascontiguousarray(d["xyz"], dtype=np.float32).reshape(-1, 3); counts = d["counts"].astype(np.int64); poses = d["poses"]
W = len(counts); fid = np.repeat(np.arange(W, dtype=np.int32), counts)
c = capi.Context(W, 0, capi.FLAG_TIMING)
for mode in ("back to back", "LM run in between", "50 ms idle in between", "fresh copies of the arrays"):
    row = []
    for rep in range(8):
        a, b = (xyz.copy(), fid.copy()) if mode.startswith("fresh") else (xyz, fid)
        c.reset_timing()
        t0 = time.perf_counter()
        c.associate(a, b, poses, 2.0, want_features=False)
        wall = (time.perf_counter() - t0) * 1e3
        row.append("%.1f/%.1f" % (c.timing()["upload"][0], wall))
        if mode.startswith("LM"):
            c.damping_iter(poses, form=0, u0=0.01, max_iter=10, min_planes=20)
        if mode.startswith("50"):
            time.sleep(0.05)
    print("%-28s upload span / call wall (ms): %s" % (mode, "  ".join(row)), flush=True)
