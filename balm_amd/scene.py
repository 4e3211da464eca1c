"""Synthetic plane-cloud scenes: the ROS-free restatement of the reference's benchmark_virtual
generator (csrc/virtual_scene.cpp; /root/reference/src/benchmark/benchmark_virtual.cpp:547-606,
:486-503).  Host-only input generation; nothing here is on the GPU hot path.  The same host library
(libbalm_scene.so) carries the readers of the shipped data formats (csrc/readers.cpp, see realworld.py).
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "lib", "libbalm_scene.so")
_SRCS = [os.path.join(_HERE, "csrc", "virtual_scene.cpp"), os.path.join(_HERE, "csrc", "readers.cpp")]
_LIB = None


def build(force=False):
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in _SRCS):
        subprocess.check_call(["g++", "-std=c++14", "-O3", "-fPIC", "-pthread", "-shared", "-o", _SO] + _SRCS)
    return _SO


def host_lib():
    return _lib()


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            build()
        _LIB = C.CDLL(_SO)
    return _LIB


@dataclass
class Scene:
    W: int
    F: int
    pts: int
    poses_gt: np.ndarray     # [W,12]  R column-major, p
    poses_init: np.ndarray   # [W,12]  ground truth + noise (benchmark_virtual.cpp:491-503)
    clusters: np.ndarray     # [F,W,10] Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz N
    coeffs: np.ndarray       # [F]
    points: np.ndarray = None  # [F,W,pts,3] float32 body-frame points (optional)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def generate(seed, W, F, pts, point_noise=0.01, surf_range=2.0, mode=0, threads=None,
             keep_points=False, feature_offset=0):
    """mode 0 = one RNG stream in the reference's draw order; mode 1 = per-feature streams,
    generated on `threads` host threads (large scenes); `feature_offset` = global index of the
    first feature (ranks of a sharded run share trajectory and pose noise, not features)."""
    if threads is None:
        threads = min(os.cpu_count() or 1, 32)
    gt = np.zeros((W, 12))
    init = np.zeros((W, 12))
    cl = np.zeros((F, W, 10))
    co = np.zeros(F)
    points = np.zeros((F, W, pts, 3), dtype=np.float32) if keep_points else None
    rc = _lib().balm_scene_generate(C.c_uint(seed), W, F, pts, C.c_double(point_noise),
                                    C.c_double(surf_range), mode, threads, int(feature_offset), _p(gt), _p(init), _p(cl),
                                    _p(co), _p(points))
    assert rc == 0
    return Scene(W, F, pts, gt, init, cl, co, points)


def sparsify(scene, seed, drop, min_obs=2):
    """Zero a random fraction of observations (sparse co-visibility); weights become sum_i N_i
    as VOX_HESS::push_voxel computes them (bavoxel.hpp:42-44)."""
    rc = _lib().balm_scene_sparsify(C.c_uint(seed), scene.W, scene.F, C.c_double(drop), min_obs,
                                    _p(scene.clusters), _p(scene.coeffs))
    assert rc == 0
    return scene
