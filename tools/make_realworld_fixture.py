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
