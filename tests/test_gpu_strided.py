"""The point-container entries of the C ABI (round 6): balm_associate_scans, balm_build_clusters_planes and
balm_window_add_scan_strided read the caller's own containers -- per-scan / per-plane arrays of 48-byte elements like the
reference's pcl::PointCloud<pcl::PointXYZINormal> (include/tools.hpp:22; benchmark_realworld.cpp:155,183-184;
benchmark_virtual.cpp:375,392-403) -- where they lie.  Index / integer work: everything they install must equal what the
flat-array entries install from the flattened copy of the same points, bit for bit."""
import os

import numpy as np
import pytest

from balm_amd import capi, scene
from conftest import ROOT
from test_association import synthetic_window
from test_gpu_voxel import cluttered_window

pytestmark = pytest.mark.gpu


def as_pcl(xyz, intensity=None, rng=None):
    """[n, 3] float32 -> [n, 12] float32 laid out like pcl::PointXYZINormal: x y z 1 | normal[4] | intensity curvature pad pad,
    with junk in every field the library must not read"""
    n = xyz.shape[0]
    rng = rng or np.random.default_rng(7)
    out = rng.standard_normal((n, 12)).astype(np.float32) * 1e6
    out[:, :3] = xyz
    out[:, 3] = 1.0
    if intensity is not None:
        out[:, 8] = intensity
    return out


@pytest.mark.parametrize("seed,W,voxel,empty_scan", [(1, 8, 1.0, None), (2, 20, 2.0, 3), (5, 9, 0.7, 0), (4, 33, 0.5, 32)])
def test_associate_scans_equals_associate_on_the_flattened_points(seed, W, voxel, empty_scan):
    poses, frames = cluttered_window(seed, W, 60, 120, 3000)
    if empty_scan is not None:
        frames[empty_scan] = frames[empty_scan][:0]
    xyz = np.concatenate(frames).astype(np.float32)
    fid = np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])
    c = capi.Context(W)
    F0, nr0, (cl0, co0, lay0, fix0, pf0) = c.associate(xyz, fid, poses, voxel_size=voxel, want_points=True)
    assert F0 > 10
    for scans in ([as_pcl(f) for f in frames], frames):                      # 48-byte elements; already packed (stride 12)
        F1, nr1, (cl1, co1, lay1, fix1, pf1) = c.associate_scans(scans, poses, voxel_size=voxel, want_points=True)
        assert (F1, nr1) == (F0, nr0)
        assert np.array_equal(cl1, cl0) and np.array_equal(co1, co0) and np.array_equal(lay1, lay0) and np.array_equal(pf1, pf0)
    c.close()


def test_associate_scans_with_marginalised_scans_and_bad_arguments():
    """the consistency driver's rules (fix_frames = 1: n_scans = win_size + 1) through the container entry"""
    from balm_amd import realworld as rw
    from test_association import exact_plane_scans
    poses, frames = exact_plane_scans(4, 9, 40, 60)
    frames = [np.asarray(f, dtype=np.float32) for f in frames]
    xyz = np.concatenate(frames).astype(np.float32)
    fid = np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])
    c = capi.Context(8)
    kw = dict(voxel_size=1.0, eigen_thresholds=rw.SIM_RULES["eigen_thresholds"], layer_limit=0, min_ps=10,
              strict=rw.SIM_RULES["strict"], fix_frames=1, min_observers=0)
    F0, _, f0 = c.associate(xyz, fid, poses, **kw)
    F1, _, f1 = c.associate_scans([as_pcl(f) for f in frames], poses, **kw)
    assert F0 == F1 and F0 >= 10
    for a, b in zip(f0, f1):
        assert (a is None and b is None) or np.array_equal(a, b)
    with pytest.raises(AssertionError):
        c.associate_scans([as_pcl(f) for f in frames[:8]], poses, **kw)        # a scan short
    import ctypes as C
    o = c._voxel_opts(1.0, (1 / 16, 1 / 16, 1 / 9), 15, 2, 2, 0, None, False)
    ptrs, cnt, stride, keep = c._containers([as_pcl(f) for f in frames[:8]])
    F, nr = C.c_int(0), C.c_long(0)
    p8 = np.ascontiguousarray(poses[:8])
    for bad_stride in (8, 46):                                                  # shorter than xyz; not a multiple of 4
        rc = c.L.balm_associate_scans(c.h, C.byref(o), 8, ptrs, cnt, bad_stride, p8.ctypes.data_as(C.c_void_p), C.byref(F), C.byref(nr))
        assert rc == capi.ERR_ARG
    rc = c.L.balm_associate_scans(c.h, C.byref(o), 7, ptrs, cnt, stride, p8.ctypes.data_as(C.c_void_p), C.byref(F), C.byref(nr))
    assert rc == capi.ERR_ARG                                                   # n_scans != win_size + fix_frames
    F2, _, _ = c.associate_scans([as_pcl(f) for f in frames[:8]], p8, voxel_size=1.0, want_features=False)   # the context is still good
    F3, _, _ = c.associate(np.concatenate(frames[:8]), np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames[:8])]), p8,
                           voxel_size=1.0, want_features=False)
    assert F2 == F3 and F2 > 0
    c.close()


@pytest.mark.parametrize("W,F,pts", [(20, 20, 40), (7, 33, 6), (64, 300, 6)])
def test_build_clusters_planes_equals_build_clusters(W, F, pts):
    """benchmark_virtual.cpp's containers: one cloud per plane, the observing pose in `intensity` (:586), pushed point by
    point into one PointCluster per (plane, pose) (:392-403)"""
    sc = scene.generate(3, W, F, pts, keep_points=True)
    pp = sc.points.reshape(F, W, pts, 3)
    planes, xyz, fid, pid = [], [], [], []
    rng = np.random.default_rng(1)
    for a in range(F):
        n_a = W * pts if a != 2 else 0                                      # one empty plane container (its clusters stay zero)
        if a == 2:
            planes.append(np.zeros((0, 12), np.float32))
            continue
        x = pp[a].reshape(-1, 3)
        inten = np.repeat(np.arange(W), pts).astype(np.float32)
        planes.append(as_pcl(x, inten, rng))
        xyz.append(x); fid.append(np.full(n_a, a, np.int32)); pid.append(inten.astype(np.int32))
    xyz, fid, pid = np.concatenate(xyz), np.concatenate(fid), np.concatenate(pid)
    fix = np.zeros((F, 10)); fix[2] = [1, 0, 0, 1, 0, 1, 0.5, 0.5, 0.5, 4]   # (the empty plane needs a point count)
    c = capi.Context(W)
    cl0 = c.build_clusters(F, xyz, fid, pid, fix, sc.coeffs)
    H0, g0, r0 = c.evaluate(0, sc.poses_init)
    cl1 = c.build_clusters_planes(planes, 8, fix, sc.coeffs)
    H1, g1, r1 = c.evaluate(0, sc.poses_init)
    assert np.array_equal(cl0, cl1)
    assert np.array_equal(H0, H1) and np.array_equal(g0, g1) and r0 == r1
    assert np.abs(cl1[2]).max() == 0 and cl1[3, :, 9].min() == pts
    c.close()


def test_window_add_scan_strided_builds_the_same_map():
    poses, frames = cluttered_window(3, 6, 40, 150, 2000)
    ca, cb = capi.Context(6), capi.Context(6)
    ca.window_open(voxel_size=1.0); cb.window_open(voxel_size=1.0)
    for i in range(6):
        ca.window_add_scan(frames[i], poses[i])
        cb.window_add_scan_strided(as_pcl(frames[i]), poses[i])
    assert ca.window_info() == cb.window_info()
    Fa, fa = ca.window_features()
    Fb, fb = cb.window_features()
    assert Fa == Fb and Fa > 5
    for a, b in zip(fa, fb):
        assert np.array_equal(a, b)
    xa, sa, _ = ca.window_points(); xb, sb, _ = cb.window_points()
    assert np.array_equal(xa, xb) and np.array_equal(sa, sb)
    ca.close(); cb.close()


def test_shipped_window_through_the_container_entry_is_the_references_feature_set():
    """the 24-scan window of datas/benchmark_realworld held as 48-byte elements -> balm_associate_scans: the feature set the
    reference's compiled cut_voxel / recut / tras_opt produced from the same files (oracle/_ref fixture), bit for bit"""
    from test_association import canon
    fx = os.path.join(ROOT, "oracle", "_ref", "realworld_scans_w24.npz")
    if not os.path.exists(fx):
        pytest.skip("fixture made from the reference's data is not present")
    d = np.load(fx)
    counts = d["counts"].astype(np.int64)
    xyz = d["xyz"].astype(np.float32)
    offs = np.concatenate([[0], np.cumsum(counts)])
    scans = [as_pcl(xyz[offs[i]:offs[i + 1]]) for i in range(len(counts))]
    c = capi.Context(len(counts))
    F, _, (cl, co, lay) = c.associate_scans(scans, d["poses"], voxel_size=2.0)
    assert F == d["clusters"].shape[0]
    assert np.array_equal(canon(cl), canon(d["clusters"]))
    assert np.array_equal(np.sort(co), np.sort(d["coeffs"]))
    c.close()


def test_cpp_shim_on_the_shipped_window_installs_the_references_feature_set(tmp_path):
    """tests/cpp/shim_realworld_e2e.cpp: the reference's translation unit + include/balm_shim.hpp on the 177-scan window held as
    pcl::PointCloud<PointXYZINormal> clouds -> BALM2_HIP::associate (balm_associate_scans) -> damping_iter.  The installed feature
    set is the one the reference's compiled cut_voxel / recut / tras_opt made from the same files, bit for bit, and the poses
    land on the reference optimizer's (1e-5 rad / 1e-4 m)."""
    from balm_amd import realworld as rw
    from test_association import canon
    ff = os.path.join(ROOT, "oracle", "_ref", "realworld_features.npz")
    if not (os.path.exists(rw.CPP_E2E_EXE) and os.path.exists(rw.SHIPPED_WINDOW_NPZ) and os.path.exists(ff)):
        pytest.skip("tools/bin/shim_realworld_e2e / datasets/realworld_w177.npz / oracle/_ref/realworld_features.npz not built")
    feats = str(tmp_path / "features.bin")
    res = rw.end_to_end_cpp(reps=2, features_out=feats)
    assert res["vs_reference"]["ok"], res
    ref = np.load(ff)
    raw = np.fromfile(feats, dtype=np.float64)
    F = int(np.frombuffer(raw[:1].tobytes(), dtype=np.int64)[0])
    W = ref["clusters"].shape[1]
    assert F == ref["clusters"].shape[0] == res["features"]
    cl = raw[1:1 + F * W * 10].reshape(F, W, 10)
    co = raw[1 + F * W * 10:]
    assert np.array_equal(canon(cl), canon(ref["clusters"]))
    assert np.array_equal(np.sort(co), np.sort(ref["coeffs"]))
