"""Shared helpers for the parity tests: scenes, tolerances, SE(3) distances."""
import numpy as np

from balm_amd import scene
from oracle import numpy_oracle as npo

# BASELINE.json north_star: final poses vs the reference CPU optimiser
ROT_TOL_RAD = 1e-5
TRANS_TOL_M = 1e-4


def make_scene(seed, W, F, pts, drop=0.0, with_fix=False, mode=0):
    sc = scene.generate(seed, W, F, pts, mode=mode)
    if drop > 0:
        scene.sparsify(sc, seed + 100, drop)
    fix = None
    if with_fix:
        # a marginalised prior per feature (benchmark_virtual.cpp:241-243): world-frame cluster of a
        # few points on the same plane = the feature's pose-0 cluster moved by the gt pose 0
        rng = np.random.default_rng(seed)
        fix = np.zeros((F, 10))
        for a in range(F):
            src = sc.clusters[a, rng.integers(0, W)]
            if src[9] > 0:
                fix[a] = src * 0.5
                fix[a, 9] = max(1.0, np.floor(src[9] * 0.5))
                fix[a, :9] = src[:9] / src[9] * fix[a, 9]
        # keep it consistent with the ground-truth world frame: transform by gt pose of that index is
        # not needed for parity (the fix cluster is just a constant term), only for realism
    return sc, fix


def pose_errors(a, b):
    """per-pose (rotation angle [rad], translation distance [m]) between two [W,12] pose arrays."""
    Ra, Rb = npo.pose_R(a), npo.pose_R(b)
    rot = np.zeros(a.shape[0])
    for i in range(a.shape[0]):
        D = Ra[i].T @ Rb[i]
        c = np.clip((np.trace(D) - 1) * 0.5, -1.0, 1.0)
        s = 0.5 * np.linalg.norm([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
        rot[i] = np.arctan2(s, c)
    tr = np.linalg.norm(npo.pose_p(a) - npo.pose_p(b), axis=1)
    return rot, tr


def rel_err(x, ref):
    return float(np.abs(np.asarray(x) - np.asarray(ref)).max() / max(np.abs(ref).max(), 1e-300))
