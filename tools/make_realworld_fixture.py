#!/usr/bin/env python3
"""Config 5 (BASELINE.json: benchmark_realworld) fixture: runs the reference's own input pipeline on the
shipped data (datas/benchmark_realworld: 177 scans, 13.4 M points) -- reader restated, association
(cut_voxel/recut/tras_opt) the reference's compiled source -- then the reference's BALM2::damping_iter,
and saves features + reference result to oracle/_ref/realworld_features.npz (git-ignored, travels to
the GPU box; ~8 MB).  Needs /root/reference (build container only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ref

src = os.environ.get("BALM_REFERENCE_ROOT", "/root/reference") + "/datas/benchmark_realworld"
t = time.time()
cl, fx, co, poses, npts = ref.realworld_features(src, 2.0)      # voxel_size 2: launch/benchmark_realworld.launch:4
t_assoc = time.time() - t
assert not (fx[:, 9] > 0).any()
t = time.time()
out, lg = ref.damping_iter(cl, None, co, poses)
t_lm = time.time() - t
dst = os.path.join(ROOT, "oracle", "_ref", "realworld_features.npz")
np.savez_compressed(dst, clusters=cl, coeffs=co, poses=poses, ref_poses=out, ref_log=lg, n_points=npts,
                    ref_seconds_lm=t_lm, ref_seconds_association=t_assoc)
nobs = (cl[..., 9] > 0).sum(1)
print("W=%d F=%d S=%d points=%d  association %.1f s, reference LM %d iterations in %.2f s -> %s (%.1f MB)"
      % (cl.shape[1], cl.shape[0], nobs.sum(), npts, t_assoc, len(lg), t_lm, dst, os.path.getsize(dst) / 1e6))
print(lg[:, :3])

# --- sub-window with its raw scans, for the device association (N3): first 24 scans, the reference's own
# cut_voxel/recut/tras_opt on them -> oracle/_ref/realworld_scans_w24.npz (~25 MB, git-ignored, travels)
import tempfile
from balm_amd import realworld as rw
WS = 24
with tempfile.TemporaryDirectory() as tmp:
    with open(os.path.join(src, "alidarPose.csv")) as f:
        lines = f.readlines()
    with open(os.path.join(tmp, "alidarPose.csv"), "w") as f:
        f.writelines(lines[:4 * WS])
    for m in range(WS):
        os.symlink(os.path.join(src, "full%d.pcd" % m), os.path.join(tmp, "full%d.pcd" % m))
    cl, fx, co, poses, npts = ref.realworld_features(tmp, 2.0)
    poses_p, frames = rw.load_window(tmp)
assert np.abs(poses_p - poses).max() < 1e-13 and sum(f.shape[0] for f in frames) == npts
dst = os.path.join(ROOT, "oracle", "_ref", "realworld_scans_w%d.npz" % WS)
np.savez_compressed(dst, xyz=np.concatenate(frames), counts=np.array([f.shape[0] for f in frames]), poses=poses,
                    clusters=cl, coeffs=co)
print("sub-window W=%d: %d points, %d features -> %s (%.1f MB)" % (WS, npts, cl.shape[0], dst, os.path.getsize(dst) / 1e6))

# --- the whole window's raw scans (13.4 M points, ~150 MB; git-ignored, travels): lets the GPU box run
# benchmark_realworld end to end (scans -> balm_associate -> LM) against the reference's features and poses
poses_p, frames = rw.load_window(src)
g = np.load(os.path.join(ROOT, "oracle", "_ref", "realworld_features.npz"))
assert np.abs(poses_p - g["poses"]).max() < 1e-13 and sum(f.shape[0] for f in frames) == int(g["n_points"])
# INPUT data (scans + initial poses) and the reference's final poses as the expected output: datasets/realworld_w177.npz is what
# bench.py's `realworld_end_to_end` leg and `python -m balm_amd.realworld --npz` read -- no checker code involved; the tests'
# older name under oracle/_ref is a link to it
os.makedirs(os.path.join(ROOT, "datasets"), exist_ok=True)
dst = os.path.join(ROOT, "datasets", "realworld_w177.npz")
np.savez_compressed(dst, xyz=np.concatenate(frames), counts=np.array([f.shape[0] for f in frames]), poses=g["poses"],
                    ref_poses=g["ref_poses"], ref_log=g["ref_log"])
lnk = os.path.join(ROOT, "oracle", "_ref", "realworld_scans_w177.npz")
if os.path.lexists(lnk):
    os.remove(lnk)
os.symlink(os.path.join("..", "..", "datasets", "realworld_w177.npz"), lnk)
print("full window: %d scans -> %s (%.1f MB)" % (len(frames), dst, os.path.getsize(dst) / 1e6))

# --- the consistency experiment's shipped scans (datas/consistency: 101 x 28 800 points, simulator output) -> N4
from balm_amd import consistency
cposes, cframes = consistency.load_window(os.environ.get("BALM_REFERENCE_ROOT", "/root/reference") + "/datas/consistency")
dst = os.path.join(ROOT, "oracle", "_ref", "consistency_scans.npz")
np.savez_compressed(dst, xyz=np.concatenate(cframes), counts=np.array([f.shape[0] for f in cframes]), poses=cposes)
print("consistency window: %d scans -> %s (%.1f MB)" % (len(cframes), dst, os.path.getsize(dst) / 1e6))
