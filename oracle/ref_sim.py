"""ctypes loader for oracle/_ref/libbalm_ref_sim.so -- the REFERENCE'S OWN consistency / covariance
sources (src/simulation/toolss.hpp, BAs_left.hpp) compiled against oracle/compat/ (ref_sim_driver.cpp,
ref_build.sh).  TEST INFRASTRUCTURE ONLY; may be absent (then `available()` is False)."""
import ctypes as C
import os

import numpy as np

from . import ref

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libbalm_ref_sim.so")
_LIB = None


def available():
    if not os.path.exists(SO) and os.path.isdir("/root/reference"):
        try:
            ref.build()
        except Exception:
            return False
    return os.path.exists(SO)


def lib():
    global _LIB
    if _LIB is None:
        if not available():
            raise ImportError("oracle/_ref/libbalm_ref_sim.so not built (needs /root/reference)")
        _LIB = C.CDLL(SO)
    return _LIB


_p, _c = ref._p, ref._c


def cluster_push(points, pn):
    pts = _c(points).reshape(-1, 3)
    cl, cc = np.zeros(10), np.zeros((9, 9))
    lib().refsim_cluster_push(_p(pts), pts.shape[0], C.c_double(pn), _p(cl), _p(cc))
    return cl, cc


def point_cov(clusters, ccov, fix, poses, beg=0, end=None):
    clusters, ccov, fix, poses = _c(clusters), _c(ccov), _c(fix), _c(poses)
    F, W = clusters.shape[:2]
    R = np.zeros((6 * W, 6 * W))
    lib().refsim_point_cov(W, F, _p(clusters), _p(ccov), _p(fix), _p(poses), beg, F if end is None else end, _p(R))
    return R.T.copy()


def pose_cov(clusters, ccov, fix, poses):
    """-> (Hess, Rcov) of the covariance tail of BALM2::damping_iter"""
    clusters, ccov, fix, poses = _c(clusters), _c(ccov), _c(fix), _c(poses)
    F, W = clusters.shape[:2]
    H, R = np.zeros((6 * W, 6 * W)), np.zeros((6 * W, 6 * W))
    lib().refsim_pose_cov(W, F, _p(clusters), _p(ccov), _p(fix), _p(poses), _p(H), _p(R))
    return H.T.copy(), R.T.copy()


def evaluate(clusters, fix, poses):
    clusters, fix, poses = _c(clusters), _c(fix), _c(poses)
    F, W = clusters.shape[:2]
    H, g = np.zeros((6 * W, 6 * W)), np.zeros(6 * W)
    r = C.c_double(0)
    lib().refsim_evaluate(W, F, _p(clusters), _p(fix), _p(poses), _p(H), _p(g), C.byref(r))
    return H.T.copy(), g, r.value


def associate(frames_xyz, poses, fix=1, voxel_size=1.0):
    """consistency.cpp:96-150 with the reference's compiled cut_voxel / recut / marginalize / tras_opt
    -> (clusters [F, W-fix, 10], fix clusters [F, 10])"""
    xyz = np.ascontiguousarray(np.concatenate(frames_xyz), dtype=np.float32)
    counts = np.array([f.shape[0] for f in frames_xyz], dtype=np.int64)
    poses = _c(poses)
    n = len(frames_xyz)
    L = lib()
    F = L.refsim_associate(n, fix, _p(xyz), _p(counts), _p(poses), C.c_double(voxel_size), None, None)
    cl, fx = np.zeros((F, n - fix, 10)), np.zeros((F, 10))
    if F:
        L.refsim_associate(n, fix, _p(xyz), _p(counts), _p(poses), C.c_double(voxel_size), _p(cl), _p(fx))
    return cl, fx


class Window:
    """BAs_left.hpp's OCTO_TREE_ROOT map driven call for call (refsim_win_*): cut_voxel / recut / marginalize / tras_opt"""

    def __init__(self, W, fix, voxel_size=1.0):
        L = lib()
        L.refsim_win_open.restype = C.c_void_p
        self.W, self.h = W, C.c_void_p(L.refsim_win_open(W, fix, C.c_double(voxel_size)))

    def cut_voxel(self, xyz, pose12):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        lib().refsim_win_cut_voxel(self.h, _p(xyz), C.c_long(xyz.shape[0]), _p(_c(pose12).reshape(12)))

    def recut(self):
        lib().refsim_win_recut(self.h)

    def marginalize(self, mg, poses=None):
        lib().refsim_win_marginalize(self.h, int(mg), _p(_c(poses)) if poses is not None else None)

    def features(self):
        F = lib().refsim_win_features(self.h)
        cl, fx = np.zeros((F, self.W, 10)), np.zeros((F, 10))
        if F:
            lib().refsim_win_export(self.h, _p(cl), _p(fx))
        return cl, fx

    def close(self):
        if self.h:
            lib().refsim_win_close(self.h)
            self.h = None
