#!/usr/bin/env python3
"""balm_window_add_scan only (a 64-scan window filled with shipped scans, nothing else): run under
`rocprofv3 --kernel-trace --stats` the kernel count / 60 is the launches per add_scan (the first four scans size buffers)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from balm_amd import capi

sdat = np.load(os.path.join(ROOT, "oracle", "_ref", "realworld_scans_w177.npz"))
poses = np.load(os.path.join(ROOT, "oracle", "_ref", "realworld_features.npz"))["poses"]
counts = sdat["counts"]
frames = np.split(sdat["xyz"], np.cumsum(counts)[:-1])
ctx = capi.Context(64)
ctx.window_open(voxel_size=2.0)
t = []
for i in range(64):
    t0 = time.perf_counter(); ctx.window_add_scan(frames[i], poses[i]); t.append(time.perf_counter() - t0)
print("64 x balm_window_add_scan: median %.3f ms, scans 4.. mean %.3f ms; %s" % (1e3 * np.median(t), 1e3 * np.mean(t[4:]), ctx.window_info()))
ctx.close()
