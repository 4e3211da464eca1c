"""ctypes loader for oracle/_ref/libbalm_ref.so -- the REFERENCE'S OWN SOURCE (tools.hpp,
bavoxel.hpp) compiled against the stand-in headers of oracle/compat/ (see ref_driver.cpp,
ref_build.sh).  TEST INFRASTRUCTURE ONLY; may be absent (then `available()` is False)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libbalm_ref.so")
_LIB = None


def build():
    subprocess.check_call(["bash", os.path.join(_HERE, "ref_build.sh")], stdout=subprocess.DEVNULL)
    return SO


def available():
    if not os.path.exists(SO) and os.path.isdir("/root/reference"):
        try:
            build()
        except Exception:
            return False
    return os.path.exists(SO)


def lib():
    global _LIB
    if _LIB is None:
        if not available():
            raise ImportError("oracle/_ref/libbalm_ref.so not built (needs /root/reference)")
        _LIB = C.CDLL(SO)
        _LIB.ref_only_residual.restype = C.c_double
        _LIB.ref_divide_thread.restype = C.c_double
        _LIB.ref_time_solve.restype = C.c_double
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def evaluate(form, clusters, fix, coeffs, poses, head=0, end=None):
    """form 0 left_evaluate_acc2, 1 acc_evaluate2, 2 left_evaluate (un-accelerated)."""
    clusters, fix, coeffs, poses = _c(clusters), _c(fix), _c(coeffs), _c(poses)
    F, W = clusters.shape[:2]
    end = F if end is None else end
    n = 6 * W
    H = np.zeros((n, n)); J = np.zeros(n); r = C.c_double(0)
    rc = lib().ref_evaluate(form, W, F, _p(clusters), _p(fix), _p(coeffs), _p(poses), head, end, _p(H), _p(J), C.byref(r))
    assert rc == 0
    return H.T.copy(), J, r.value


def only_residual(clusters, fix, coeffs, poses):
    clusters, fix, coeffs, poses = _c(clusters), _c(fix), _c(coeffs), _c(poses)
    F, W = clusters.shape[:2]
    return lib().ref_only_residual(W, F, _p(clusters), _p(fix), _p(coeffs), _p(poses))


def divide_thread(form, clusters, fix, coeffs, poses):
    clusters, fix, coeffs, poses = _c(clusters), _c(fix), _c(coeffs), _c(poses)
    F, W = clusters.shape[:2]
    n = 6 * W
    H = np.zeros((n, n)); J = np.zeros(n)
    r = lib().ref_divide_thread(form, W, F, _p(clusters), _p(fix), _p(coeffs), _p(poses), _p(H), _p(J))
    return H.T.copy(), J, r


def solve_damped(H, g, u):
    Hc = _c(np.asarray(H).T); g = _c(g)
    n = g.shape[0]
    dx = np.zeros(n); q1 = C.c_double(0)
    lib().ref_solve_damped(n, _p(Hc), _p(g), C.c_double(u), _p(dx), C.byref(q1))
    return dx, q1.value


def damping_iter(clusters, fix, coeffs, poses):
    """BALM2::damping_iter (left form, u0 = 0.01, <= 10 iterations) -> (poses, log[rows, 8])."""
    clusters, fix, coeffs = _c(clusters), _c(fix), _c(coeffs)
    out = _c(poses).copy()
    F, W = clusters.shape[:2]
    lg = np.zeros((16, 8))
    rows = lib().ref_damping_iter(W, F, _p(clusters), _p(fix), _p(coeffs), _p(out), _p(lg), 16)
    return out, lg[:rows].copy()


def push_voxel(clusters_a):
    clusters_a = _c(clusters_a)
    coe = C.c_double(0)
    kept = lib().ref_push_voxel(clusters_a.shape[0], _p(clusters_a), C.byref(coe))
    return bool(kept), coe.value


def time_sample(clusters, coeffs, poses, f_sample):
    clusters, coeffs, poses = _c(clusters), _c(coeffs), _c(poses)
    out = np.zeros(2)
    lib().ref_time_sample(clusters.shape[1], f_sample, _p(clusters), _p(coeffs), _p(poses), _p(out))
    return float(out[0]), float(out[1])


SO_V3 = os.path.join(_HERE, "_ref", "libbalm_ref_v3.so")      # the same sources built with -march=x86-64-v3 (ref_build.sh)
_LIB_V3 = None


def v3_available():
    """the AVX2/FMA build exists and this host can run it"""
    if not os.path.exists(SO_V3):
        return False
    try:
        flags = open("/proc/cpuinfo").read().split("flags", 1)[1].split("\n", 1)[0].split()
    except Exception:
        return False
    return all(f in flags for f in ("avx2", "fma", "bmi2", "movbe", "f16c"))


def time_sample_v3(clusters, coeffs, poses, f_sample):
    global _LIB_V3
    if _LIB_V3 is None:
        _LIB_V3 = C.CDLL(SO_V3, mode=os.RTLD_LOCAL)
        _LIB_V3.ref_time_solve.restype = C.c_double
    clusters, coeffs, poses = _c(clusters), _c(coeffs), _c(poses)
    out = np.zeros(2)
    _LIB_V3.ref_time_sample(clusters.shape[1], f_sample, _p(clusters), _p(coeffs), _p(poses), _p(out))
    return float(out[0]), float(out[1])


def time_solve_v3(H, g, u):
    Hc = _c(np.asarray(H).T); g = _c(g)
    return _LIB_V3.ref_time_solve(g.shape[0], _p(Hc), _p(g), C.c_double(u))


def time_solve(H, g, u):
    Hc = _c(np.asarray(H).T); g = _c(g)
    return lib().ref_time_solve(g.shape[0], _p(Hc), _p(g), C.c_double(u))


def exp(w):
    R = np.zeros(9)
    lib().ref_exp(_p(_c(w)), _p(R))
    return R.reshape(3, 3).T.copy()


def log(R):
    w = np.zeros(3)
    lib().ref_log(_p(_c(np.asarray(R).T)), _p(w))
    return w


def realworld_features(data_dir, voxel_size=2.0, max_poses=0):
    """benchmark_realworld's input pipeline (ROS-free): read the shipped alidarPose.csv + full<m>.pcd, run
    the reference's own cut_voxel / recut / tras_opt, return (clusters [F,W,10], fix [F,10], coeffs [F],
    poses [W,12], n_points)."""
    L = lib()
    L.ref_rw_open.restype = C.c_void_p
    h = L.ref_rw_open(data_dir.encode(), C.c_double(voxel_size), int(max_poses))
    if not h:
        raise FileNotFoundError(data_dir)
    h = C.c_void_p(h)
    W, F, npts = C.c_int(0), C.c_int(0), C.c_long(0)
    L.ref_rw_dims(h, C.byref(W), C.byref(F), C.byref(npts))
    cl = np.zeros((F.value, W.value, 10)); fx = np.zeros((F.value, 10)); co = np.zeros(F.value)
    poses = np.zeros((W.value, 12))
    L.ref_rw_export(h, _p(cl), _p(fx), _p(co), _p(poses))
    L.ref_rw_close(h)
    return cl, fx, co, poses, npts.value


class Window:
    """the reference's octree used incrementally (ref_driver.cpp: cut_voxel + recut per scan, marginalize, tras_opt)"""

    def __init__(self, W, voxel_size=2.0, eigen_thresholds=(1.0 / 16, 1.0 / 16, 1.0 / 9), min_ps=15, layer_limit=2):
        L = lib()
        L.ref_win_open.restype = C.c_void_p
        thr = (C.c_float * 3)(*[float(t) for t in eigen_thresholds])
        self.W = W
        self.h = C.c_void_p(L.ref_win_open(int(W), C.c_double(voxel_size), thr, int(min_ps), int(layer_limit)))

    def add_scan(self, xyz, pose12):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        lib().ref_win_add_scan(self.h, _p(xyz), C.c_long(xyz.shape[0]), _p(_c(pose12)))

    def marginalize(self, mg, poses=None):
        lib().ref_win_marginalize(self.h, int(mg), _p(_c(poses)))

    def features(self):
        """-> clusters [F,W,10], fix [F,10], coeffs [F]"""
        F = lib().ref_win_features(self.h)
        cl, fix, co = np.zeros((F, self.W, 10)), np.zeros((F, 10)), np.zeros(F)
        if F:
            lib().ref_win_export(self.h, _p(cl), _p(fix), _p(co))
        return cl, fix, co

    def close(self):
        if self.h:
            lib().ref_win_close(self.h)
            self.h = None
