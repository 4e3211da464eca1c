#!/usr/bin/env python3
"""Sliding-window map timing on the shipped scans (oracle/_ref/realworld_scans_w177.npz: test infrastructure output, used here
only as INPUT data): per call wall time of balm_window_add_scan / _features / _marginalize.
   python tools/bench_window.py [--window 20 --slide 5 --scans 60]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from balm_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--window", type=int, default=20)
ap.add_argument("--slide", type=int, default=5)
ap.add_argument("--scans", type=int, default=60)
a = ap.parse_args()
sdat = np.load(os.path.join(ROOT, "oracle", "_ref", "realworld_scans_w177.npz"))
poses = np.load(os.path.join(ROOT, "oracle", "_ref", "realworld_features.npz"))["poses"]
counts = sdat["counts"]
frames = np.split(sdat["xyz"], np.cumsum(counts)[:-1])
W = a.window
ctx = capi.Context(W, 0, capi.FLAG_TIMING)
ctx.window_open(voxel_size=2.0)
t_add, t_feat, t_marg, t_lm = [], [], [], []
inwin = []
for i in range(a.scans):
    t0 = time.perf_counter(); ctx.window_add_scan(frames[i], poses[i]); t_add.append(time.perf_counter() - t0)
    inwin.append(poses[i])
    if len(inwin) == W:
        t0 = time.perf_counter(); F, _ = ctx.window_features(want_features=False); t_feat.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); out, lg = ctx.damping_iter(np.stack(inwin), form=0, u0=0.01, max_iter=10, reanchor=False); t_lm.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); ctx.window_marginalize(a.slide, out); t_marg.append(time.perf_counter() - t0)
        inwin = list(out[a.slide:])
ms, n = ctx.timing()["voxel"]
scans, pts, nodes = ctx.window_info()
med = lambda v: 1e3 * float(np.median(v))       # medians: the first calls of each kind size the session's buffers
print("window %d, slide %d, %d scans of ~%d points, median call times: add_scan %.2f ms (with the window full: %.2f), features %.2f ms, "
      "LM %.2f ms (%d its last), marginalize %.2f ms; device time in balm_window_* %.2f ms per call; %d points, %d nodes resident"
      % (W, a.slide, a.scans, int(counts[:a.scans].mean()), med(t_add), med(t_add[W:]), med(t_feat), med(t_lm), len(lg), med(t_marg),
         ms / max(n, 1), pts, nodes))
ctx.close()
