"""ctypes loader for the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product package ``balm_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("balm_oracle_capi.cpp", "balm_oracle.hpp")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_only_residual.restype = C.c_double
        _LIB.orc_evaluate_threads.restype = C.c_double
        _LIB.orc_time_solve.restype = C.c_double
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def exp(w):
    R = np.zeros(9)
    lib().orc_exp(_p(_c(w)), _p(R))
    return R.reshape(3, 3).T.copy()


def log(R):
    w = np.zeros(3)
    lib().orc_log(_p(_c(np.asarray(R).T)), _p(w))
    return w


def eig3(A):
    lam = np.zeros(3)
    U = np.zeros(9)
    lib().orc_eig3(_p(_c(np.asarray(A).T)), _p(lam), _p(U))
    return lam, U.reshape(3, 3).T.copy()


def cluster_push(xyz):
    xyz = _c(xyz).reshape(-1, 3)
    cl = np.zeros(10)
    lib().orc_cluster_push(_p(xyz), C.c_long(xyz.shape[0]), _p(cl))
    return cl


def evaluate(form, clusters, fix, coeffs, poses, head=0, end=None):
    """-> (Hess [n,n], JacT [n], residual).  clusters [F,W,10], poses [W,12]."""
    clusters, fix, coeffs, poses = _c(clusters), _c(fix), _c(coeffs), _c(poses)
    F, W = clusters.shape[0], clusters.shape[1]
    end = F if end is None else end
    n = 6 * W
    H = np.zeros((n, n))
    J = np.zeros(n)
    r = C.c_double(0)
    rc = lib().orc_evaluate(form, W, F, _p(clusters), _p(fix), _p(coeffs), _p(poses), head, end,
                            _p(H), _p(J), C.byref(r))
    assert rc == 0, rc
    return H.T.copy(), J, r.value   # column-major -> numpy [r,c]


def evaluate_threads(form, clusters, fix, coeffs, poses, threads):
    clusters, fix, coeffs, poses = _c(clusters), _c(fix), _c(coeffs), _c(poses)
    F, W = clusters.shape[0], clusters.shape[1]
    n = 6 * W
    H = np.zeros((n, n))
    J = np.zeros(n)
    r = lib().orc_evaluate_threads(form, W, F, _p(clusters), _p(fix), _p(coeffs), _p(poses),
                                   threads, _p(H), _p(J))
    return H.T.copy(), J, r


def only_residual(clusters, fix, coeffs, poses):
    clusters, fix, coeffs, poses = _c(clusters), _c(fix), _c(coeffs), _c(poses)
    F, W = clusters.shape[0], clusters.shape[1]
    return lib().orc_only_residual(W, F, _p(clusters), _p(fix), _p(coeffs), _p(poses))


def ldlt_solve(A, b):
    A = _c(np.asarray(A).T)   # column-major
    b = _c(b)
    n = b.shape[0]
    x = np.zeros(n)
    neg = lib().orc_ldlt_solve(n, _p(A), _p(b), _p(x))
    return x, neg


def solve_damped(H, g, u):
    Hc = _c(np.asarray(H).T)
    g = _c(g)
    n = g.shape[0]
    dx = np.zeros(n)
    q1 = C.c_double(0)
    lib().orc_solve_damped(n, _p(Hc), _p(g), C.c_double(u), _p(dx), C.byref(q1))
    return dx, q1.value


def update_poses(form, poses, dxi):
    poses, dxi = _c(poses), _c(dxi)
    out = np.zeros_like(poses)
    lib().orc_update_poses(form, poses.shape[0], _p(poses), _p(dxi), _p(out))
    return out


def reanchor(poses):
    out = _c(poses).copy()
    lib().orc_reanchor(out.shape[0], _p(out))
    return out


def damping_iter(form, clusters, fix, coeffs, poses, u0, max_iter, rel_tol=1e-6, threads=1):
    """-> (poses_out [W,12], log [iters,8]: r1 r2 u v q q1 accepted hess_evaluated)."""
    clusters, fix, coeffs = _c(clusters), _c(fix), _c(coeffs)
    out = _c(poses).copy()
    F, W = clusters.shape[0], clusters.shape[1]
    lg = np.zeros((max_iter, 8))
    it = lib().orc_damping_iter(form, W, F, _p(clusters), _p(fix), _p(coeffs), _p(out),
                                C.c_double(u0), max_iter, C.c_double(rel_tol), threads, _p(lg))
    return out, lg[:it].copy()


def rsme(gt, es):
    gt, es = _c(gt), _c(es)
    r, t = C.c_double(0), C.c_double(0)
    lib().orc_rsme(gt.shape[0], _p(gt), _p(es), C.byref(r), C.byref(t))
    return r.value, t.value


def time_sample(form, clusters, fix, coeffs, poses, f_sample, threads):
    """seconds for (one Hessian evaluation, one residual-only evaluation) on features [0,f_sample)."""
    clusters, fix, coeffs, poses = _c(clusters), _c(fix), _c(coeffs), _c(poses)
    W = clusters.shape[1]
    out = np.zeros(2)
    lib().orc_time_sample(form, W, f_sample, _p(clusters), _p(fix), _p(coeffs), _p(poses), threads,
                          _p(out))
    return float(out[0]), float(out[1])


def time_solve(H, g, u):
    Hc = _c(np.asarray(H).T)
    g = _c(g)
    return lib().orc_time_solve(g.shape[0], _p(Hc), _p(g), C.c_double(u))
