#!/usr/bin/env python3
"""Where does the pinned-ring upload lose its rate inside a Python process?  balm_set_features (800 MB) and the shipped window's
balm_associate upload, (a) as is, (b) with the process pinned to one block of 64 hardware threads before the library's pool starts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] != "none":
    lo, hi = [int(x) for x in sys.argv[1].split("-")]
    os.sched_setaffinity(0, set(range(lo, hi + 1)))
print("affinity: %d cpus (%s)" % (len(os.sched_getaffinity(0)), sys.argv[1] if len(sys.argv) > 1 else "none"), flush=True)
from balm_amd import capi, scene
W, F = 200, 30000
sc = scene.generate(2024, W, F, 6, mode=1)
c = capi.Context(W, 0, capi.FLAG_TIMING)
for rep in range(4):
    c.reset_timing()
    t0 = time.perf_counter()
    c.set_features(sc.clusters, None, sc.coeffs)
    up = c.timing()["upload"][0]
    print("set_features %.0f MB rep %d: wall %.2f ms, upload span %.2f ms = %.1f GB/s" % (sc.clusters.nbytes / 1e6, rep, (time.perf_counter() - t0) * 1e3, up, sc.clusters.nbytes / up / 1e6), flush=True)
c.close()
