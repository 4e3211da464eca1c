#!/usr/bin/env python3
"""Golden vectors for the association (N2/N3), produced by the reference's OWN compiled state machines:
  assoc_bench_w8.npz : cut_voxel / recut / tras_opt of src/benchmark/bavoxel.hpp (through oracle/_ref/libbalm_ref.so's
                       restated file reader, on scans written in the shipped formats), voxel_size 1
  assoc_sim_w8.npz   : the copy in src/simulation/BAs_left.hpp with consistency.cpp's flow (first scan marginalised into
                       fix clusters, strict plane test, layer_limit 0), voxel_size 1
Each file holds the scans (float32), the poses, and the reference's feature set.  Needs /root/reference."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import ref, ref_sim
from test_association import exact_plane_scans, write_window
from test_gpu_voxel import cluttered_window

dst = os.path.join(ROOT, "tests", "golden")
# benchmark rules: cluttered scans so that all three octree layers produce features
poses, frames = cluttered_window(5, 8, 30, 90, 900)
with tempfile.TemporaryDirectory() as tmp:
    write_window(tmp, poses, frames)
    cl, fx, co, poses_r, npts = ref.realworld_features(tmp, 1.0)
    from balm_amd import realworld as rw
    poses_p, frames_p = rw.load_window(tmp)                 # what the file round trip (%.9f poses) hands the reference
assert npts == sum(f.shape[0] for f in frames_p) and not (fx[:, 9] > 0).any()
np.savez_compressed(os.path.join(dst, "assoc_bench_w8.npz"), xyz=np.concatenate(frames_p),
                    counts=np.array([f.shape[0] for f in frames_p]), poses=poses_r, clusters=cl, coeffs=co)
print("bench rules: %d points, %d features" % (npts, cl.shape[0]))
# the consistency driver's rules: exact planes, 9 scans of which the first is marginalised
poses, frames = exact_plane_scans(6, 9, 30, 50)
cl, fx = ref_sim.associate(frames, poses, 1, 1.0)
np.savez_compressed(os.path.join(dst, "assoc_sim_w8.npz"), xyz=np.concatenate(frames), counts=np.array([f.shape[0] for f in frames]),
                    poses=poses, clusters=cl, fix=fx)
print("consistency rules: %d points, %d features, %d with a fix cluster" % (sum(f.shape[0] for f in frames), cl.shape[0], (fx[:, 9] > 0).sum()))
for f in ("assoc_bench_w8.npz", "assoc_sim_w8.npz"):
    print(f, os.path.getsize(os.path.join(dst, f)), "bytes")
