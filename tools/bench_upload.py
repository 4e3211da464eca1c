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


if "--shards" in sys.argv:
    # The set-up of a one-process multi-device context (BALM_FLAG_LOOPBACK_SHARDS: all shards on this box's one GPU, i.e. ONE link):
    # balm_set_features / _cb / balm_build_clusters of BASELINE configs[3]'s table cut into N shards against the same bytes through one
    # context.  Per byte the sharded set-up should cost about what the single upload costs (the link is shared; nothing is serialised
    # behind one host pool, nothing is built on device 0 and shipped back).
    N = int(sys.argv[sys.argv.index("--shards") + 1])
    Wm, Fm = 200, int(os.environ.get("F", 100000))                 # 1.6 GB of clusters (configs[3] is 3.2 GB: two of these)
    scm = scene.generate(7, Wm, Fm, 6, mode=1, keep_points=True)
    feat = np.repeat(np.arange(Fm, dtype=np.int32), Wm * 6)
    pose = np.tile(np.repeat(np.arange(Wm, dtype=np.int32), 6), Fm)
    xyz = np.ascontiguousarray(scm.points.reshape(-1, 3), dtype=np.float32)
    rows = [scm.clusters[f] for f in range(Fm)]
    res = {}
    for n in (1, N):
        c = capi.Context(Wm, 0, capi.FLAG_TIMING | (capi.FLAG_LOOPBACK_SHARDS if n > 1 else 0), n_devices=n if n > 1 else None)
        for name, fn, nbytes in (("balm_set_features", lambda: c.set_features(scm.clusters, None, scm.coeffs), scm.clusters.nbytes),
                                 ("balm_build_clusters", lambda: c.build_clusters(Fm, xyz, feat, pose, None, scm.coeffs, want_clusters=False),
                                  xyz.nbytes + feat.nbytes + pose.nbytes)):
            ts = []
            for rep in range(4):
                c.reset_timing()
                t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
            res[(name, n)] = (min(ts[1:]), nbytes)
            if n > 1:
                tm = [c.shard_timing(k) for k in range(n)]
                print("      last call %.1f ms; per shard: upload span %s ms; build kernel %s ms" %
                      (ts[-1], " ".join("%.1f" % t["upload"][0] for t in tm), " ".join("%.2f" % t["build"][0] for t in tm)))
            print("%-22s %d shard%s  %8.1f MB  wall %8.2f ms (best of 3 warm) = %5.1f GB/s" % (name, n, " " if n == 1 else "s", nbytes / 1e6, min(ts[1:]), nbytes / min(ts[1:]) / 1e6), flush=True)
        c.close()
    for name in ("balm_set_features", "balm_build_clusters"):
        print("%s: %d shards / 1 shard = %.2f x per byte" % (name, N, res[(name, N)][0] / res[(name, 1)][0]))
    sys.exit(0)

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
