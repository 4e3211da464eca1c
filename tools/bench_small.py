#!/usr/bin/env python3
"""LM iterations/s of small windows (BASELINE configs[0], configs[1] and the shipped window's shape) with and without
the hipGraph replay of the iteration."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balm_amd import capi, scene
for W, F, pts in ((20, 20, 40), (20, 150, 40), (64, 5000, 6), (100, 2000, 6)):
    sc = scene.generate(1, W, F, pts, mode=1)
    row = []
    for graph in (False, True):
        if graph:
            os.environ["BALM_GRAPH"] = "1"
        else:
            os.environ.pop("BALM_GRAPH", None)
        c = capi.Context(W)
        c.set_features(sc.clusters, None, sc.coeffs)
        c.damping_iter(sc.poses_init, u0=0.1, max_iter=20, force_hess=True, no_stop=True, reanchor=False)
        t0 = time.perf_counter()
        K, reps = 20, 10
        for _ in range(reps):
            c.damping_iter(sc.poses_init, u0=0.1, max_iter=K, force_hess=True, no_stop=True, reanchor=False)
        row.append((time.perf_counter() - t0) / (K * reps) * 1e3)
        c.close()
    print("W=%4d F=%6d: %.3f ms/step plain launches, %.3f ms/step graph replay (x%.2f)" % (W, F, row[0], row[1], row[0] / row[1]), flush=True)
