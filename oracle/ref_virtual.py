"""ctypes loader for oracle/_ref/libbalm_ref_virtual.so -- the REFERENCE'S OWN benchmark_virtual.cpp
(its copy of class BALM2) compiled against oracle/compat/ (see ref_virtual_driver.cpp, ref_build.sh).
TEST INFRASTRUCTURE ONLY; may be absent (then `available()` is False)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libbalm_ref_virtual.so")
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
            raise ImportError("oracle/_ref/libbalm_ref_virtual.so not built (needs /root/reference)")
        _LIB = C.CDLL(SO)
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def damping_iter(points, poses):
    """BALM2::dampingIter(x_stats, plSurfs) of benchmark_virtual.cpp:375-482.
    points [F,W,pts,3] float32 body-frame, poses [W,12] -> (poses_out, log [it,8], seconds)."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    F, W, pts = points.shape[:3]
    out = _c(poses).copy()
    lg = np.zeros((20, 8))
    sec = C.c_double(0)
    rows = lib().refv_damping_iter(W, F, pts, _p(points), _p(out), _p(lg), 20, C.byref(sec))
    return out, lg[:rows].copy(), sec.value


def evaluate(form, clusters, fix, coeffs, poses):
    """form 0 left_evaluate_acc2, 1 accEvaluate2, 3 only_residual (H, J untouched) of the virtual copy."""
    clusters, fix, coeffs, poses = _c(clusters), _c(fix), _c(coeffs), _c(poses)
    F, W = clusters.shape[:2]
    n = 6 * W
    H = np.zeros((n, n)); J = np.zeros(n); r = C.c_double(0)
    rc = lib().refv_evaluate(form, W, F, _p(clusters), _p(fix), _p(coeffs), _p(poses), _p(H), _p(J), C.byref(r))
    assert rc == 0
    return H.T.copy(), J, r.value


def rsme(gt, es):
    gt, es = _c(gt), _c(es)
    r, t = C.c_double(0), C.c_double(0)
    lib().refv_rsme(gt.shape[0], _p(gt), _p(es), C.byref(r), C.byref(t))
    return r.value, t.value
