#!/usr/bin/env python3
"""What the drop-in's caller pays in front of the device work: the C ABI's big uploads from PAGEABLE host memory.
  balm_set_features   W=200 / F=50 000 (800 MB cluster table, BASELINE configs[2]) -- flat array and the fill-callback form
  balm_build_clusters 24 M points (12 + 4 + 4 bytes each)
  balm_associate      the shipped 177-scan window (161 MB of points + 54 MB of scan ids), if datasets/realworld_w177.npz travels
Per call: wall ms, the BALM_T_UPLOAD span (first DMA start -> last DMA end on the stream, host fills included) and its GB/s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from balm_amd import capi, scene, realworld as rw


def line(name, nbytes, wall_ms, t):
    up = t["upload"][0] / max(t["upload"][1], 1)
    print("%-44s %8.1f MB  wall %8.2f ms  upload span %8.2f ms = %6.1f GB/s  (%4.1f GB/s over the whole call)"
          % (name, nbytes / 1e6, wall_ms, up, nbytes / up / 1e6 if up > 0 else 0, nbytes / wall_ms / 1e6), flush=True)


W, F = 200, int(os.environ.get("F", 50000))
sc = scene.generate(2024, W, F, 6, mode=1)
c = capi.Context(W, 0, capi.FLAG_TIMING)
for rep in range(4):
    c.reset_timing()
    t0 = time.perf_counter()
    c.set_features(sc.clusters, None, sc.coeffs)
    line("balm_set_features W=200 F=%d rep %d" % (F, rep), sc.clusters.nbytes, (time.perf_counter() - t0) * 1e3, c.timing())
H, g, r = c.evaluate(0, sc.poses_init)
print("  residual after the staged upload: %.9g" % r)
c.close()

# N1: points -> clusters
Fb, pts = 20000, 6
scb = scene.generate(5, W, Fb, pts, mode=1, keep_points=True)
feat = np.repeat(np.arange(Fb, dtype=np.int32), W * pts)
pose = np.tile(np.repeat(np.arange(W, dtype=np.int32), pts), Fb)
xyz = np.ascontiguousarray(scb.points.reshape(-1, 3), dtype=np.float32)
c = capi.Context(W, 0, capi.FLAG_TIMING)
for rep in range(3):
    c.reset_timing()
    t0 = time.perf_counter()
    c.build_clusters(Fb, xyz, feat, pose, None, scb.coeffs, want_clusters=False)
    t = c.timing()
    line("balm_build_clusters %d M points rep %d" % (xyz.shape[0] // 1000000, rep), xyz.nbytes + feat.nbytes + pose.nbytes,
         (time.perf_counter() - t0) * 1e3, t)
    print("    build kernel %.3f ms" % t["build"][0])
c.close()

if os.path.exists(rw.SHIPPED_WINDOW_NPZ):
    import json
    res = rw.end_to_end(rw.SHIPPED_WINDOW_NPZ, 0, reps=5)
    print("shipped window end to end:", json.dumps(res))
