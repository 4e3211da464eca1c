#!/usr/bin/env python3
"""N1 measurement: GPU cluster build (k_build_clusters) from raw points, HBM roofline.
   python tools/bench_cluster_build.py [--win 200 --features 20000 --pts 6]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from balm_amd import capi, scene

ap = argparse.ArgumentParser()
ap.add_argument("--win", type=int, default=200)
ap.add_argument("--features", type=int, default=20000)
ap.add_argument("--pts", type=int, default=6)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
W, F, pts = a.win, a.features, a.pts
sc = scene.generate(11, W, F, pts, mode=1, keep_points=True)
xyz = sc.points.reshape(-1, 3)
fid = np.repeat(np.arange(F, dtype=np.int32), W * pts)
pid = np.tile(np.repeat(np.arange(W, dtype=np.int32), pts), F)
ctx = capi.Context(W, 0, capi.FLAG_TIMING)
ctx.build_clusters(F, xyz, fid, pid, None, sc.coeffs, want_clusters=False)     # warm-up
ctx.reset_timing()
t0 = time.perf_counter()
for _ in range(a.reps):
    ctx.build_clusters(F, xyz, fid, pid, None, sc.coeffs, want_clusters=False)
wall = (time.perf_counter() - t0) / a.reps
ms, n = ctx.timing()["build"]
n_pts = xyz.shape[0]
alg_bytes = 20.0 * n_pts + 80.0 * F * W            # 12 B xyz + 8 B keys per point in, one 80-B cluster per (a,i) out
kern = ms / n * 1e-3
got = ctx.build_clusters(F, xyz, fid, pid, None, sc.coeffs)
err = float(np.abs(got - sc.clusters).max() / np.abs(sc.clusters).max())
print(json.dumps({"kernel": "k_build_clusters", "points": n_pts, "W": W, "F": F, "kernel_ms": kern * 1e3,
                  "algorithmic_bytes": alg_bytes, "achieved_GBps": alg_bytes / kern / 1e9, "peak_GBps": 8000.0, "deterministic": True, "atomics": 0,
                  "frac_of_8TBps": alg_bytes / kern / 8e12, "points_per_s": n_pts / kern,
                  "call_wall_ms_incl_pcie": wall * 1e3, "max_rel_err_vs_host_push": err}))
