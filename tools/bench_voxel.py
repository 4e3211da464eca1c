#!/usr/bin/env python3
"""N3 measurement: balm_associate (device) vs the host association on a synthetic window of the shipped
data's size (W=177 scans x ~76 k points = 13.4 M points).  Prints device ms (HIP events, kernels only),
wall ms through the C ABI (incl. PCIe upload of 161 MB of points), host C++ seconds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from balm_amd import capi, realworld as rw
from oracle import assoc_host as ah
from test_gpu_voxel import cluttered_window

W = int(os.environ.get("W", 177))
real = os.path.join(ROOT, "oracle", "_ref", "realworld_scans_w177.npz")
if "--real" in sys.argv and os.path.exists(real):
    d = np.load(real)
    frames = np.split(d["xyz"], np.cumsum(d["counts"])[:-1])
    poses = np.load(os.path.join(ROOT, "oracle", "_ref", "realworld_features.npz"))["poses"]
    W = len(frames)
else:
    poses, frames = cluttered_window(7, W, 400, 150, 16000)
npts = sum(f.shape[0] for f in frames)
c = capi.Context(W, flags=capi.FLAG_TIMING)
xyz = np.concatenate(frames); fid = np.concatenate([np.full(f.shape[0], i, np.int32) for i, f in enumerate(frames)])
for rep in range(3):
    c.reset_timing()
    t = time.time()
    F, nroots, _ = c.associate(xyz, fid, poses, 2.0, want_features=False)
    wall = (time.time() - t) * 1e3
    ms = c.timing()["voxel"][0]
print("points %d  roots %d  features %d  device %.2f ms (%.2f Gpoints/s)  wall %.1f ms" % (npts, nroots, F, ms, npts / ms / 1e6, wall))
if "--no-cpu" not in sys.argv:
    t = time.time()
    cl, co, _ = ah.associate(frames, poses, 2.0)
    print("host C++ association: %d features in %.2f s" % (cl.shape[0], time.time() - t))
