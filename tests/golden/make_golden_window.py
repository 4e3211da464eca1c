#!/usr/bin/env python3
"""Golden feature tables of the reference's octree used INCREMENTALLY (the comparator sequence of balm_window_*): made by the
reference's own OCTO_TREE_ROOT compiled from src/benchmark/bavoxel.hpp (oracle/ref_driver.cpp ref_win_*: cut_voxel + recut
per scan, tras_opt, marginalize with poses), so that the GPU test has something to compare with where oracle/_ref is absent.

    python tests/golden/make_golden_window.py          # needs /root/reference (build container only)

Inputs regenerate from the seed (tests/test_gpu_voxel.py::cluttered_window + the pose perturbations below); the fixture holds,
per snapshot (window full, then after every slide), the canonically ordered per-scan clusters and fix clusters of the feature
table (float64, exact)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref  # noqa: E402
from oracle import numpy_oracle as npo  # noqa: E402

SEED, W, MG, SLIDES = 3, 8, 2, 2


def noisy(poses, rng, s_rot, s_tr):
    out = poses.copy()
    for i in range(poses.shape[0]):
        R = poses[i, :9].reshape(3, 3).T
        out[i, :9] = (npo.exp_so3(s_rot * rng.standard_normal(3)) @ R).T.reshape(9)
        out[i, 9:] = poses[i, 9:] + s_tr * rng.standard_normal(3)
    return out


def sequence():
    """yields ("add", scan index, pose) / ("marg", mg, poses) / ("snap", tag) -- the same calls for both sides"""
    from test_gpu_voxel import cluttered_window
    total = W + SLIDES * MG
    poses, frames = cluttered_window(SEED, total, 40, 120, 1500)
    rng = np.random.default_rng(SEED)
    start = noisy(poses, rng, 2e-3, 2e-2)
    for i in range(W):
        yield ("add", frames[i], start[i])
    yield ("snap", "full")
    nxt = W
    for sl in range(SLIDES):
        yield ("marg", MG, noisy(poses[nxt - W:nxt], rng, 2e-4, 2e-3))
        for _ in range(MG):
            yield ("add", frames[nxt], start[nxt])
            nxt += 1
        yield ("snap", "slide%d" % sl)


def canon(cl, fix):
    o = np.lexsort(cl.reshape(cl.shape[0], -1)[:, ::-1].T)
    return cl[o], fix[o]


if __name__ == "__main__":
    win = ref.Window(W, voxel_size=1.0)
    out = {}
    for step in sequence():
        if step[0] == "add":
            win.add_scan(step[1], step[2])
        elif step[0] == "marg":
            win.marginalize(step[1], step[2])
        else:
            cl, fix, co = win.features()
            cl, fix = canon(cl, fix)
            out["cl_" + step[1]], out["fix_" + step[1]] = cl, fix
            print(step[1], cl.shape, int((fix[:, 9] > 0).sum()), "with a fix cluster")
    win.close()
    np.savez_compressed(os.path.join(HERE, "window_w8_mg2.npz"), **out)
