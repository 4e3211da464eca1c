"""ctypes binding of the C ABI in include/balm_hip.h (balm_amd/lib/libbalm_hip.so).

This is the only way Python reaches the HIP path; there is no CPU fallback.  Loading fails loudly
when the library has not been built (``python -m balm_amd.build``).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BALM_HIP_LIB") or os.path.join(_HERE, "lib", "libbalm_hip.so")   # override: A/B builds

FORM_LEFT, FORM_RIGHT = 0, 1
OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_TOO_FEW_PLANES, ERR_NUMERIC = range(6)
ABI_VERSION = 6            # include/balm_hip.h: BALM_ABI_VERSION
FLAG_TIMING = 1
FLAG_LOOPBACK_SHARDS = 2
FLAG_SYRK_INT8 = 4        # opt-in: dense Gt Gt^T products on the INT8 matrix cores; default FP64
T_MOMENTS, T_FACTORS, T_SYRK, T_ASSEMBLE, T_SOLVE, T_UPDATE, T_BUILD, T_VOXEL, T_COV, T_COMM, T_UPLOAD, T_COUNT = range(12)
TIMING_NAMES = ["moments", "factors", "syrk", "assemble", "solve", "update", "build", "voxel", "cov", "comm", "upload"]

# every symbol include/balm_hip.h declares
EXPORTS = ["balm_create", "balm_prewarm", "balm_create_multi", "balm_destroy", "balm_set_features", "balm_set_features_cb", "balm_evaluate", "balm_only_residual",
           "balm_solve_damped", "balm_damping_iter", "balm_build_clusters", "balm_build_clusters_planes", "balm_voxel_defaults", "balm_associate", "balm_associate_scans", "balm_get_features", "balm_get_association", "balm_pose_covariance",
           "balm_window_open", "balm_window_add_scan", "balm_window_add_scan_strided", "balm_window_recut", "balm_window_get_points", "balm_window_features", "balm_window_marginalize", "balm_window_info", "balm_window_close",
           "balm_set_allreduce", "balm_comm_unique_id", "balm_comm_init_rank", "balm_comm_info",
           "balm_get_timing", "balm_get_shard_timing", "balm_get_solve_trace", "balm_chain_macro_plan", "balm_reset_timing", "balm_work_model", "balm_last_error", "balm_version", "balm_abi_version"]


class IterLog(C.Structure):
    _fields_ = [("r1", C.c_double), ("r2", C.c_double), ("u", C.c_double), ("v", C.c_double),
                ("q", C.c_double), ("q1", C.c_double), ("accepted", C.c_int), ("hess_evaluated", C.c_int)]


class LMOpts(C.Structure):
    _fields_ = [("form", C.c_int), ("u0", C.c_double), ("max_iter", C.c_int), ("rel_tol", C.c_double),
                ("min_planes_per_pose", C.c_int), ("force_hess", C.c_int), ("no_stop", C.c_int),
                ("verbose", C.c_int), ("reanchor", C.c_int), ("abs_tol", C.c_double)]


class VoxelOpts(C.Structure):
    _fields_ = [("voxel_size", C.c_double), ("eigen_thr", C.c_float * 3), ("min_ps", C.c_int), ("layer_limit", C.c_int),
                ("min_observers", C.c_int), ("fix_frames", C.c_int), ("max_plane_dist", C.c_double),
                ("max_lambda21", C.c_double), ("max_lambda0", C.c_double), ("want_point_features", C.c_int),
                ("fix_point_limit", C.c_int), ("defer_recut", C.c_int)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_long, C.c_void_p)
FILL_CLUSTERS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double))

_LIB = None


class BalmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("balm_hip error %d: %s" % (code, msg))
        self.code = code


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libbalm_hip.so is not built (%s missing): run `python -m balm_amd.build`. "
                              "There is no CPU fallback for the HIP path." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        abi = getattr(L, "balm_abi_version", None)      # (a library older than the symbol itself: revision 0)
        abi = abi() if abi is not None else 0
        if abi != ABI_VERSION:
            raise ImportError("libbalm_hip.so has ABI revision %d, this binding was written for %d: rebuild (python -m balm_amd.build)"
                              % (abi, ABI_VERSION))
        L.balm_create.restype = C.c_void_p
        L.balm_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.balm_create_multi.restype = C.c_void_p
        L.balm_create_multi.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.balm_prewarm.argtypes = [C.c_int]
        L.balm_destroy.restype = None
        L.balm_destroy.argtypes = [C.c_void_p]
        L.balm_set_features.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.balm_set_features_cb.argtypes = [C.c_void_p, C.c_int, FILL_CLUSTERS_FN, C.c_void_p, C.c_void_p, C.c_void_p]
        L.balm_evaluate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_double)]
        L.balm_only_residual.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.balm_solve_damped.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                        C.POINTER(C.c_double)]
        L.balm_damping_iter.argtypes = [C.c_void_p, C.POINTER(LMOpts), C.c_void_p, C.POINTER(IterLog),
                                        C.POINTER(C.c_int)]
        L.balm_build_clusters.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.balm_build_clusters_planes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]
        L.balm_associate_scans.argtypes = [C.c_void_p, C.POINTER(VoxelOpts), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_long)]
        L.balm_window_add_scan_strided.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_size_t, C.c_void_p]
        L.balm_window_open.argtypes = [C.c_void_p, C.POINTER(VoxelOpts)]
        L.balm_window_add_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        L.balm_window_recut.argtypes = [C.c_void_p]
        L.balm_window_get_points.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        L.balm_window_features.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.balm_window_marginalize.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.balm_window_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.balm_window_close.argtypes = [C.c_void_p]
        L.balm_associate.argtypes = [C.c_void_p, C.POINTER(VoxelOpts), C.c_void_p, C.c_void_p, C.c_long, C.c_void_p,
                                     C.POINTER(C.c_int), C.POINTER(C.c_long)]
        L.balm_get_features.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.balm_get_association.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.balm_voxel_defaults.restype = None
        L.balm_voxel_defaults.argtypes = [C.POINTER(VoxelOpts)]
        L.balm_pose_covariance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        L.balm_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
        L.balm_comm_unique_id.argtypes = [C.c_void_p]
        L.balm_comm_init_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.balm_comm_info.argtypes = [C.c_void_p, C.c_void_p]
        L.balm_get_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.balm_get_shard_timing.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.balm_get_solve_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        L.balm_chain_macro_plan.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_long]
        L.balm_reset_timing.argtypes = [C.c_void_p]
        L.balm_work_model.argtypes = [C.c_void_p, C.c_void_p]
        L.balm_last_error.restype = C.c_char_p
        L.balm_last_error.argtypes = [C.c_void_p]
        L.balm_version.restype = C.c_char_p
        _LIB = L
    return _LIB


def chain_macro_plan(panels, helpers):
    """host-only: the macro-tiles every helper workgroup of k_ldl_chain owns -> list (per helper) of (r0, j0)"""
    tab = np.empty(helpers * 64, np.int32)
    rc = lib().balm_chain_macro_plan(int(panels), int(helpers), tab.ctypes.data_as(C.c_void_p), tab.size)
    if rc != OK:
        raise BalmError(rc, "balm_chain_macro_plan(%d, %d)" % (panels, helpers))
    tab = tab.reshape(helpers, 64)
    return [[(int(e) & 0xffff, int(e) >> 16) for e in row if e >= 0] for row in tab]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype=np.float64):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


class Context:
    """One balm_ctx: owns the HBM-resident problem for one GPU."""

    def __init__(self, win_size, device=0, flags=0, n_devices=None):
        """n_devices=None: balm_create (one GPU, no collective path); n_devices >= 1: balm_create_multi over
        GPUs device .. device+n_devices-1 of this process (features sharded, RCCL all-reduce inside the library)."""
        self.L = lib()
        self.W = int(win_size)
        self.n = 6 * self.W
        if n_devices is None:
            self.h = self.L.balm_create(self.W, int(device), int(flags))
        else:
            self.h = self.L.balm_create_multi(self.W, int(device), int(n_devices), int(flags))
        if not self.h:
            raise BalmError(ERR_HIP, "balm_create(win_size=%d, device=%d, n_devices=%s) failed (no GPU, bad device, or "
                                     "win_size out of range)" % (win_size, device, n_devices))
        self.F = 0
        self._cb = None

    def close(self):
        if getattr(self, "h", None):
            self.L.balm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise BalmError(rc, self.L.balm_last_error(self.h).decode())

    def set_features(self, clusters, fix, coeffs):
        clusters, fix, coeffs = _c(clusters), _c(fix), _c(coeffs)
        F = clusters.shape[0]
        assert clusters.shape == (F, self.W, 10), clusters.shape
        assert coeffs.shape == (F,)
        assert fix is None or fix.shape == (F, 10)
        self._check(self.L.balm_set_features(self.h, F, _p(clusters), _p(fix), _p(coeffs)))
        self.F = F

    def set_features_cb(self, per_feature, fix, coeffs):
        """balm_set_features_cb: `per_feature` is a sequence of F separate [W, 10] arrays (the shape of VOX_HESS's borrowed
        `vector<PointCluster>*`); the library's host threads pull feature ranges straight into its pinned staging chunks."""
        tabs = [_c(t) for t in per_feature]
        F = len(tabs)
        assert all(t.shape == (self.W, 10) for t in tabs)
        fix, coeffs = _c(fix), _c(coeffs)
        row = self.W * 10

        failed = []

        def fill(_user, f0, f1, dst):
            # ctypes prints and swallows an exception raised in a callback: the chunk would leave with stale bytes and the
            # call would still answer BALM_OK.  Zero the range (an all-zero cluster = "not observed"), remember, re-raise after.
            out = np.ctypeslib.as_array(dst, shape=((f1 - f0) * row,))
            try:
                for a in range(f0, f1):
                    out[(a - f0) * row:(a - f0 + 1) * row] = tabs[a].reshape(-1)
            except BaseException as exc:       # noqa: BLE001
                out[:] = 0.0
                failed.append(exc)

        cb = FILL_CLUSTERS_FN(fill)
        rc = self.L.balm_set_features_cb(self.h, F, cb, None, _p(fix), _p(coeffs))
        if failed:
            self.F = 0
            raise failed[0]
        self._check(rc)
        self.F = F

    def build_clusters(self, F, xyz, feat_id, pose_id, fix, coeffs, want_clusters=True):
        xyz = _c(xyz, np.float32).reshape(-1, 3)
        feat_id, pose_id = _c(feat_id, np.int32), _c(pose_id, np.int32)
        fix, coeffs = _c(fix), _c(coeffs)
        out = np.zeros((F, self.W, 10)) if want_clusters else None
        self._check(self.L.balm_build_clusters(self.h, F, _p(xyz), _p(feat_id), _p(pose_id), xyz.shape[0],
                                               _p(fix), _p(coeffs), _p(out)))
        self.F = F
        return out

    @staticmethod
    def _containers(arrays):
        """list of C-contiguous float32 [n_k, c] arrays (c >= 3: x, y, z lead each row, like the 12 floats of a
        pcl::PointXYZINormal) -> (pointer array, count array, stride in bytes, the arrays kept alive)"""
        keep = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]
        keep = [a.reshape(-1, 3) if a.ndim == 1 else a for a in keep]
        cols = {a.shape[1] for a in keep}
        assert len(cols) == 1 and min(cols) >= 3, "every container: [n, c] float32 with the same c >= 3"
        ptrs = (C.c_void_p * len(keep))(*[a.ctypes.data if a.shape[0] else None for a in keep])
        cnt = (C.c_long * len(keep))(*[a.shape[0] for a in keep])
        return ptrs, cnt, 4 * cols.pop(), keep

    def build_clusters_planes(self, planes, pose_col, fix, coeffs, want_clusters=True):
        """balm_build_clusters_planes: `planes` = F separate [n_a, c] float32 arrays (one per plane, as benchmark_virtual.cpp
        holds its clouds), column `pose_col` of each row = the observing pose stored as a float (`intensity`)"""
        ptrs, cnt, stride, keep = self._containers(planes)
        F = len(keep)
        fix, coeffs = _c(fix), _c(coeffs)
        out = np.zeros((F, self.W, 10)) if want_clusters else None
        self._check(self.L.balm_build_clusters_planes(self.h, F, ptrs, cnt, stride, 4 * int(pose_col), _p(fix), _p(coeffs), _p(out)))
        self.F = F
        return out

    def associate_scans(self, scans, poses, voxel_size=2.0, eigen_thresholds=(1.0 / 16, 1.0 / 16, 1.0 / 9), min_ps=15,
                        want_features=True, layer_limit=2, min_observers=2, fix_frames=0, strict=None, want_points=False):
        """balm_associate_scans: `scans` = one [n_i, c] float32 array per scan (c = 3: packed xyz; c = 12: 48-byte elements
        like pcl::PointXYZINormal), read by the library where they lie.  Returns what associate() returns."""
        ptrs, cnt, stride, keep = self._containers(scans)
        poses = _c(poses)
        assert poses.shape[0] == self.W + fix_frames == len(keep), "one scan and one pose per window slot"
        o = self._voxel_opts(voxel_size, eigen_thresholds, min_ps, layer_limit, min_observers, fix_frames, strict, want_points)
        F, nr = C.c_int(0), C.c_long(0)
        self._check(self.L.balm_associate_scans(self.h, C.byref(o), len(keep), ptrs, cnt, stride, _p(poses), C.byref(F), C.byref(nr)))
        self.F = F.value
        return self.F, nr.value, self._association_result(want_features, fix_frames, want_points, sum(a.shape[0] for a in keep))

    def _voxel_opts(self, voxel_size, eigen_thresholds, min_ps, layer_limit, min_observers, fix_frames, strict, want_points):
        o = VoxelOpts()
        self.L.balm_voxel_defaults(C.byref(o))
        o.voxel_size = voxel_size
        o.eigen_thr = (C.c_float * 3)(*[float(t) for t in eigen_thresholds])
        o.min_ps, o.layer_limit, o.min_observers, o.fix_frames = min_ps, layer_limit, min_observers, fix_frames
        if strict is not None:
            o.max_plane_dist, o.max_lambda21, o.max_lambda0 = strict
        o.want_point_features = int(want_points)
        return o

    def _association_result(self, want_features, fix_frames, want_points, n_pts):
        feats = None
        if want_features and self.F > 0:
            cl, co = np.zeros((self.F, self.W, 10)), np.zeros(self.F)
            layer = np.zeros(self.F, dtype=np.int32)
            self._check(self.L.balm_get_features(self.h, _p(cl), _p(co), _p(layer)))
            feats = (cl, co, layer)
            if fix_frames or want_points:
                fix = np.zeros((self.F, 10))
                pf = np.zeros(n_pts, dtype=np.int32) if want_points else None
                self._check(self.L.balm_get_association(self.h, _p(fix), _p(pf)))
                feats = (cl, co, layer, fix, pf)
        return feats

    def associate(self, xyz, frame_id, poses, voxel_size=2.0, eigen_thresholds=(1.0 / 16, 1.0 / 16, 1.0 / 9),
                  min_ps=15, want_features=True, layer_limit=2, min_observers=2, fix_frames=0, strict=None,
                  want_points=False):
        """Adaptive-voxel association on the device; installs the features.  -> (F, n_root_voxels,
        (clusters [F,W,10], coeffs [F], layer [F]) or None); with fix_frames / want_points the tuple continues
        with fix [F,10] and the feature index of every point [n] (-1 = none)"""
        xyz = _c(xyz, np.float32).reshape(-1, 3)
        frame_id, poses = _c(frame_id, np.int32), _c(poses)
        assert poses.shape[0] == self.W + fix_frames, "poses for the marginalised scans too"
        o = self._voxel_opts(voxel_size, eigen_thresholds, min_ps, layer_limit, min_observers, fix_frames, strict, want_points)
        F, nr = C.c_int(0), C.c_long(0)
        self._check(self.L.balm_associate(self.h, C.byref(o), _p(xyz), _p(frame_id), xyz.shape[0], _p(poses),
                                          C.byref(F), C.byref(nr)))
        self.F = F.value
        return self.F, nr.value, self._association_result(want_features, fix_frames, want_points, xyz.shape[0])

    # ---- sliding-window map (the incremental use of the reference's octree) ----
    def window_open(self, voxel_size=2.0, eigen_thresholds=(1.0 / 16, 1.0 / 16, 1.0 / 9), min_ps=15, layer_limit=2,
                    min_observers=2, fix_frames=0, strict=None, fix_point_limit=50, defer_recut=False):
        """strict = (max_plane_dist, max_lambda21, max_lambda0): the consistency driver's plane test (BAs_left.hpp:674)"""
        o = VoxelOpts()
        self.L.balm_voxel_defaults(C.byref(o))
        o.voxel_size = voxel_size
        o.eigen_thr = (C.c_float * 3)(*[float(t) for t in eigen_thresholds])
        o.min_ps, o.layer_limit, o.min_observers = min_ps, layer_limit, min_observers
        o.fix_frames, o.fix_point_limit, o.defer_recut = int(fix_frames), int(fix_point_limit), int(bool(defer_recut))
        if strict is not None:
            o.max_plane_dist, o.max_lambda21, o.max_lambda0 = [float(v) for v in strict]
        self._check(self.L.balm_window_open(self.h, C.byref(o)))

    def window_points(self):
        """-> (xyz [n,3] float32 body frame, window slot [n], feature of the last window_features [n] or -1), scan order"""
        n = self.window_info()[1]
        xyz, slot, feat = np.zeros((n, 3), np.float32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        got = C.c_long(0)
        self._check(self.L.balm_window_get_points(self.h, _p(xyz), _p(slot), _p(feat), n, C.byref(got)))
        return xyz[:got.value], slot[:got.value], feat[:got.value]

    def window_recut(self):
        """OCTO_TREE_ROOT::recut over the scans no recut has seen (window_open(defer_recut=True): consistency.cpp:127-136)"""
        self._check(self.L.balm_window_recut(self.h))

    def window_add_scan(self, xyz, pose12):
        """cut_voxel + recut: body-frame points of one scan, its pose [12]"""
        xyz, pose12 = _c(xyz, np.float32).reshape(-1, 3), _c(pose12).reshape(12)
        self._check(self.L.balm_window_add_scan(self.h, _p(xyz), xyz.shape[0], _p(pose12)))

    def window_add_scan_strided(self, points, pose12):
        """the same for a [n, c] float32 container whose rows start with x, y, z (c = 12: pcl::PointXYZINormal elements)"""
        ptrs, cnt, stride, keep = self._containers([points])
        pose12 = _c(pose12).reshape(12)
        self._check(self.L.balm_window_add_scan_strided(self.h, keep[0].ctypes.data, keep[0].shape[0], stride, _p(pose12)))

    def window_marginalize(self, mg_size, poses=None):
        """OCTO_TREE_ROOT::marginalize of every root; poses [scans_in_window, 12] (re-transform) or None"""
        if poses is not None:
            poses = _c(poses)
            assert poses.shape[0] == self.window_info()[0]
        self._check(self.L.balm_window_marginalize(self.h, int(mg_size), _p(poses)))

    def window_features(self, want_features=True):
        """tras_opt: installs the window's feature table -> (F, (clusters [F,W,10], coeffs [F], layer [F], fix [F,10]))"""
        F = C.c_int(0)
        self._check(self.L.balm_window_features(self.h, C.byref(F)))
        self.F = F.value
        if not (want_features and self.F > 0):
            return self.F, None
        cl, co = np.zeros((self.F, self.W, 10)), np.zeros(self.F)
        layer, fix = np.zeros(self.F, dtype=np.int32), np.zeros((self.F, 10))
        self._check(self.L.balm_get_features(self.h, _p(cl), _p(co), _p(layer)))
        self._check(self.L.balm_get_association(self.h, _p(fix), None))
        return self.F, (cl, co, layer, fix)

    def window_info(self):
        a, b, c = C.c_int(0), C.c_long(0), C.c_long(0)
        self._check(self.L.balm_window_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def window_close(self):
        self._check(self.L.balm_window_close(self.h))

    def pose_covariance(self, poses, cluster_cov=None, point_sigma=0.0, want_raw=True):
        """-> (Rcov [n,n], Rcov_raw [n,n] or None): H^-1 Rcov_raw H^-T and the point-noise image sum Ls c_cov Ls^T"""
        poses = _c(poses)
        cc = None
        if cluster_cov is not None:
            cc = _c(cluster_cov).reshape(self.F, self.W, 81)
        R = np.zeros((self.n, self.n))
        Rraw = np.zeros((self.n, self.n)) if want_raw else None
        self._check(self.L.balm_pose_covariance(self.h, _p(poses), _p(cc), float(point_sigma), _p(R), _p(Rraw)))
        return R, Rraw

    def evaluate(self, form, poses, head=0, end=None, want_hess=True):
        """-> (Hess [n,n] or None, JacT [n], residual)"""
        poses = _c(poses)
        end = self.F if end is None else end
        H = np.zeros((self.n, self.n)) if want_hess else None
        g = np.zeros(self.n)
        r = C.c_double(0)
        self._check(self.L.balm_evaluate(self.h, form, _p(poses), head, end, _p(H), _p(g), C.byref(r)))
        if H is not None:
            H = H.T.copy()    # column-major -> numpy [row, col]
        return H, g, r.value

    def only_residual(self, poses):
        poses = _c(poses)
        r = C.c_double(0)
        self._check(self.L.balm_only_residual(self.h, _p(poses), C.byref(r)))
        return r.value

    def solve_damped(self, H, g, u):
        Hc = _c(np.asarray(H).T)
        g = _c(g)
        dx = np.zeros(self.n)
        q1 = C.c_double(0)
        self._check(self.L.balm_solve_damped(self.h, _p(Hc), _p(g), u, _p(dx), C.byref(q1)))
        return dx, q1.value

    def damping_iter(self, poses, form=FORM_LEFT, u0=0.01, max_iter=10, rel_tol=1e-6, min_planes=0,
                     force_hess=False, no_stop=False, verbose=False, reanchor=True, abs_tol=0.0):
        """-> (poses_out [W,12], log [iters, 8]: r1 r2 u v q q1 accepted hess_evaluated)"""
        out = _c(poses).copy()
        o = LMOpts(form, u0, max_iter, rel_tol, min_planes, int(force_hess), int(no_stop), int(verbose),
                   int(reanchor), abs_tol)
        lg = (IterLog * max_iter)()
        it = C.c_int(0)
        self._check(self.L.balm_damping_iter(self.h, C.byref(o), _p(out), lg, C.byref(it)))
        log = np.array([[e.r1, e.r2, e.u, e.v, e.q, e.q1, e.accepted, e.hess_evaluated]
                        for e in lg[:it.value]]).reshape(-1, 8)
        return out, log

    def set_allreduce(self, fn):
        """fn(dev_ptr: int, n_doubles: int) -> None; sums the device buffer across ranks in place."""
        if fn is None:
            self._cb = None
            self._check(self.L.balm_set_allreduce(self.h, C.cast(None, ALLREDUCE_FN), None))
            return

        def tramp(ptr, n, _user):
            try:
                fn(ptr, n)
                return 0
            except Exception as e:   # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        self._cb = ALLREDUCE_FN(tramp)
        self._check(self.L.balm_set_allreduce(self.h, self._cb, None))

    @staticmethod
    def comm_unique_id():
        """128 bytes (ncclUniqueId) drawn by ONE rank; hand them to every rank's comm_init_rank."""
        buf = C.create_string_buffer(128)
        rc = lib().balm_comm_unique_id(buf)
        if rc != OK:
            raise BalmError(rc, "balm_comm_unique_id failed (librccl.so.1 not loadable?)")
        return buf.raw

    def comm_init_rank(self, n_ranks, rank, unique_id):
        """RCCL inside the library for the one-process-per-GPU launch: no hook, no host synchronisation."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self.L.balm_comm_init_rank(self.h, int(n_ranks), int(rank), buf))

    def comm_info(self):
        """what the transport itself reports: ranks (ncclCommCount), own rank, payload doubles per evaluation, kind"""
        out = (C.c_long * 4)()
        self._check(self.L.balm_comm_info(self.h, out))
        return {"ranks": int(out[0]), "rank": int(out[1]), "payload_doubles": int(out[2]),
                "transport": ["none", "rccl-in-library", "loopback-shards", "caller-hook"][int(out[3])]}

    def timing(self):
        ms = np.zeros(T_COUNT)
        cnt = np.zeros(T_COUNT, dtype=np.int64)
        self._check(self.L.balm_get_timing(self.h, _p(ms), _p(cnt)))
        return {TIMING_NAMES[k]: (float(ms[k]), int(cnt[k])) for k in range(T_COUNT)}

    def shard_timing(self, shard):
        """balm_get_shard_timing: the timers of ONE device of a multi-device context"""
        ms = np.zeros(T_COUNT)
        cnt = np.zeros(T_COUNT, dtype=np.int64)
        self._check(self.L.balm_get_shard_timing(self.h, int(shard), _p(ms), _p(cnt)))
        return {TIMING_NAMES[k]: (float(ms[k]), int(cnt[k])) for k in range(T_COUNT)}

    def solve_trace(self):
        """[2P+1, P, 6] wall-clock ticks (100 MHz) of the last persistent factorisation (needs BALM_SOLVE_TRACE=1)"""
        dims = (C.c_int * 3)()
        self.L.balm_get_solve_trace(self.h, None, 0, dims)
        n6 = dims[0] * dims[1] * dims[2]
        out = np.zeros(n6 + 6 * dims[1], dtype=np.int64)
        self._check(self.L.balm_get_solve_trace(self.h, _p(out), out.size, dims))
        self.step_phases = out[n6:].reshape(dims[1], 6)      # per column: sums over the 12 steps of publish+barrier | pivot+rows | barrier | operands+MFMA
        return out[:n6].reshape(dims[0], dims[1], dims[2])

    def reset_timing(self):
        self._check(self.L.balm_reset_timing(self.h))

    def work_model(self):
        out = np.zeros(4)
        self._check(self.L.balm_work_model(self.h, _p(out)))
        return {"S": out[0], "B": out[1], "syrk_flops_algorithmic": out[2], "syrk_flops_issued": out[3]}
