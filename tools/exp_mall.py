#!/usr/bin/env python3
"""Experiment: does k_feature_factors (K2) of a feature range run faster when k_world_moments (K1) has just streamed the
same clusters (Infinity Cache reuse)?  Sub-range evaluations run K1 -> K1b -> K2 back to back on the range."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balm_amd import capi, scene
W, F = 200, 50000
sc = scene.generate(2024, W, F, 6, mode=1)
c = capi.Context(W, 0, capi.FLAG_TIMING)
c.set_features(sc.clusters, None, sc.coeffs)
for chunk in (50000, 25000, 12500, 6250, 3125):
    c.evaluate(0, sc.poses_init, 0, chunk, want_hess=False)
    c.reset_timing()
    reps = 0
    for lo in range(0, F, chunk):
        c.evaluate(0, sc.poses_init, lo, lo + chunk, want_hess=False)
        reps += 1
    t = c.timing()
    print("chunk %6d features (%4.0f MB of clusters): moments %.3f ms, factors %.3f ms per 50k features  (%d launches)"
          % (chunk, chunk * W * 80 / 1e6, t["moments"][0], t["factors"][0], reps), flush=True)
