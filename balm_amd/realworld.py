"""ROS/PCL-free real-world pipeline: what the reference's benchmark_realworld driver
does around the optimizer (src/benchmark/benchmark_realworld.cpp:144-218) -- read alidarPose.csv and the
binary PCD scans, express poses relative to pose 0, associate points to plane features by adaptive
voxelisation (csrc/association.cpp restating bavoxel.hpp's cut_voxel / recut / tras_opt) -- feeding the
GPU path through the C ABI.  `associate` is the host C++ association; `associate_gpu` (--gpu-assoc) runs the
same decisions on the device (balm_associate, csrc/kernels_voxel.hip).

    python -m balm_amd.realworld /path/to/datas/benchmark_realworld [--voxel 2.0] [--gpu-assoc]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

from . import scene


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _lib():
    L = scene.host_lib()
    L.balm_assoc_create.restype = C.c_void_p
    L.balm_read_pcd_xyz.restype = C.c_long
    return L


def read_pose_csv(path, max_poses=100000):
    buf = np.zeros((max_poses, 12))
    stamps = np.zeros(max_poses)
    n = _lib().balm_read_pose_csv(path.encode(), max_poses, _p(buf), _p(stamps))
    if n < 0:
        raise FileNotFoundError(path)
    return buf[:n].copy(), stamps[:n].copy()


def read_pcd_xyz(path):
    L = _lib()
    n = L.balm_read_pcd_xyz(path.encode(), None, C.c_long(0))
    if n < 0:
        raise IOError("cannot read binary PCD %s (%d)" % (path, n))
    xyz = np.zeros((n, 3), dtype=np.float32)
    got = L.balm_read_pcd_xyz(path.encode(), _p(xyz), C.c_long(n))
    return xyz[:got]


def relative_to_first(poses):
    """benchmark_realworld.cpp:163-168: p_i <- R_0^T (p_i - p_0), R_i <- R_0^T R_i"""
    R = poses[:, :9].reshape(-1, 3, 3).transpose(0, 2, 1)
    p = poses[:, 9:]
    Rn = np.einsum("ji,njk->nik", R[0], R)
    pn = (p - p[0]) @ R[0]
    out = np.zeros_like(poses)
    out[:, :9] = Rn.transpose(0, 2, 1).reshape(-1, 9)
    out[:, 9:] = pn
    return out


# the consistency driver's association rules: src/simulation/BAs_left.hpp:18-23 (layer_limit 0, thresholds 1/64,
# min_ps 10), :674 (max point-to-plane distance 1 mm, lambda2/lambda1 < 25, lambda0 < 1e-10), consistency.cpp:125-131
# (first scan marginalised into fix clusters), BAs_left.hpp:38 (no observer minimum)
SIM_RULES = dict(voxel_size=1.0, eigen_thresholds=(1.0 / 64, 1.0 / 64, 1.0 / 64), layer_limit=0, min_ps=10,
                 strict=(0.001, 25.0, 1e-10), fix_frames=1, min_observers=0)


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


def associate_gpu(ctx, frames_xyz, poses, voxel_size=2.0, eigen_thresholds=(1.0 / 16, 1.0 / 16, 1.0 / 9), layer_limit=2,
                  min_ps=15, strict=None, fix_frames=0, min_observers=2, want_points=False, want_features=True):
    """Same contract and rule set as `associate` (e.g. `**SIM_RULES`), on the device through balm_associate; the
    features are installed in `ctx` (a context for len(frames_xyz) - fix_frames poses).  -> (F, n_root_voxels,
    feature tuple or None); the point -> feature map refers to the concatenation of the scans."""
    xyz = np.concatenate([np.ascontiguousarray(f, dtype=np.float32).reshape(-1, 3) for f in frames_xyz])
    fid = np.concatenate([np.full(f.shape[0], i, dtype=np.int32) for i, f in enumerate(frames_xyz)])
    return ctx.associate(xyz, fid, poses, voxel_size, eigen_thresholds, min_ps, want_features, layer_limit, min_observers,
                         fix_frames, strict, want_points)


def canonical_order(clusters, coeffs):
    """Order-independent comparison key for feature sets (hash-map order on the host, key order on the device)."""
    first = np.argmax(clusters[:, :, 9] > 0, axis=1)
    head = clusters[np.arange(clusters.shape[0]), first]
    return np.lexsort((head[:, 8], head[:, 7], head[:, 6], first, coeffs))


def load_window(data_dir, max_poses=None):
    poses, stamps = read_pose_csv(os.path.join(data_dir, "alidarPose.csv"))
    if max_poses:
        poses = poses[:max_poses]
    frames = [read_pcd_xyz(os.path.join(data_dir, "full%d.pcd" % m)) for m in range(poses.shape[0])]
    return relative_to_first(poses), frames


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("data_dir")
    ap.add_argument("--voxel", type=float, default=2.0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--gpu-assoc", action="store_true", help="adaptive voxelisation on the GPU (balm_associate)")
    ap.add_argument("--out", default=None, help="write optimised poses (W x 12) here as .npy")
    a = ap.parse_args(argv)
    from . import capi
    t = time.time()
    poses, frames = load_window(a.data_dir)
    t_read = time.time() - t
    W = poses.shape[0]
    ctx = capi.Context(W, a.device)
    t = time.time()
    if a.gpu_assoc:
        F = associate_gpu(ctx, frames, poses, a.voxel, want_features=False)[0]
    else:
        cl, co, _ = associate(frames, poses, a.voxel)
        F = cl.shape[0]
    t_assoc = time.time() - t
    print("The size of poses: %d" % W)                                   # benchmark_realworld.cpp:171
    print("read %d points in %.1f s; %d plane features in %.1f s" % (sum(f.shape[0] for f in frames), t_read, F, t_assoc))
    if F < 3 * W:                                                        # :209-215
        print("Initial error too large.\nPlease loose plane determination criteria for more planes.\n"
              "The optimization is terminated.")
        return 1
    if not a.gpu_assoc:
        ctx.set_features(cl, None, co)
    t = time.time()
    out, lg = ctx.damping_iter(poses, form=capi.FORM_LEFT, u0=0.01, max_iter=10, min_planes=20, verbose=True)
    print("optimised in %d LM iterations, %.2f ms" % (len(lg), (time.time() - t) * 1e3))
    if a.out:
        np.save(a.out, out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
