#!/usr/bin/env python3
"""LM iteration of a 20-pose (or W-pose) window over a range of feature counts: python tools/bench_w20.py [W]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balm_amd import capi, scene
W = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for F in (20, 150, 1000, 3000, 10000, 50000):
    sc = scene.generate(1, W, F, 6, mode=1)
    c = capi.Context(W)
    c.set_features(sc.clusters, None, sc.coeffs)
    c.damping_iter(sc.poses_init, u0=0.1, max_iter=20, force_hess=True, no_stop=True, reanchor=False)
    t0 = time.perf_counter()
    K, reps = 20, 10
    for _ in range(reps):
        c.damping_iter(sc.poses_init, u0=0.1, max_iter=K, force_hess=True, no_stop=True, reanchor=False)
    print("W=%4d F=%6d: %.3f ms/step" % (W, F, (time.perf_counter() - t0) / (K * reps) * 1e3), flush=True)
    c.close()
