"""ctypes loader for oracle/libassoc_host.so -- the host restatement of the reference's adaptive-voxel association
(oracle/host_association.cpp), the comparator of the device association (balm_associate).  TEST INFRASTRUCTURE ONLY:
only tests/ and tools/ may import this module; the product package associates on the device."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libassoc_host.so")
_LIB = None


def build(force=False):
    src = os.path.join(_HERE, "host_association.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(src) > os.path.getmtime(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libassoc_host.so"], stdout=subprocess.DEVNULL)
    return _SO


def _lib():
    global _LIB
    if _LIB is None:
        build()
        _LIB = C.CDLL(_SO)
        _LIB.balm_assoc_create.restype = C.c_void_p
        _LIB.balm_assoc_export_points.restype = C.c_long
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def associate(frames_xyz, poses, voxel_size=2.0, eigen_thresholds=(1.0 / 16, 1.0 / 16, 1.0 / 9), layer_limit=2,
              min_ps=15, strict=None, fix_frames=0, min_observers=2, want_points=False):
    """frames_xyz: list of [n_i,3] float32 body-frame scans; poses [W,12].  Returns (clusters [F,W,10],
    coeffs [F], layer [F]).  Defaults = benchmark_realworld.cpp:183-185 + launch/benchmark_realworld.launch:4.
    With fix_frames / strict / want_points (the consistency driver, `**SIM_RULES`): W counts the scans after the
    marginalised ones and the result is (clusters, coeffs, layer, fix [F,10], points) with points = (xyz [n,3]
    float32, feature [n], scan [n]) of every feature, or None."""
    L = _lib()
    L.balm_assoc_export_points.restype = C.c_long
    W = len(frames_xyz)
    thr = np.asarray(eigen_thresholds, dtype=np.float32)
    h = C.c_void_p(L.balm_assoc_create(W, C.c_double(voxel_size), _p(thr), layer_limit, min_ps))
    extended = strict is not None or fix_frames or want_points or min_observers != 2
    try:
        if extended:
            st = strict or (0.0, 0.0, 0.0)
            L.balm_assoc_set_rules(h, C.c_double(st[0]), C.c_double(st[1]), C.c_double(st[2]), fix_frames, min_observers)
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        for i, xyz in enumerate(frames_xyz):
            xyz = np.ascontiguousarray(xyz, dtype=np.float32)
            rc = L.balm_assoc_add_frame(h, i, _p(xyz), C.c_long(xyz.shape[0]), _p(poses[i]))
            assert rc == 0
        F = L.balm_assoc_finish(h)
        cl = np.zeros((F, W - fix_frames, 10))
        co = np.zeros(F)
        layer = np.zeros(F, dtype=np.int32)
        L.balm_assoc_export(h, _p(cl), _p(co), _p(layer))
        if extended:
            fix = np.zeros((F, 10))
            L.balm_assoc_export_fix(h, _p(fix))
            pts = None
            if want_points:
                n = L.balm_assoc_export_points(h, None, None, None)
                xyz, fid, sid = np.zeros((n, 3), np.float32), np.zeros(n, np.int32), np.zeros(n, np.int32)
                L.balm_assoc_export_points(h, _p(xyz), _p(fid), _p(sid))
                pts = (xyz, fid, sid)
    finally:
        L.balm_assoc_destroy(h)
    if extended:
        return cl, co, layer, fix, pts
    return cl, co, layer


