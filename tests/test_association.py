"""N2: the product's ROS-free real-world input pipeline -- pose CSV / binary PCD readers
and the adaptive-voxel association (oracle/host_association.cpp) -- against the reference's own
cut_voxel / recut / tras_opt compiled in oracle/_ref.  Integer/index work: bit-exact (as feature sets;
the reference's feature ORDER is its unordered_map's iteration order)."""
import os

import numpy as np
import pytest

from balm_amd import realworld as rw
from oracle import assoc_host as ah
from balm_amd import scene
from oracle import numpy_oracle as npo
from oracle import ref

needs_ref = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")


def canon(cl):
    """order-free canonical form of a feature set [F,W,10]"""
    flat = cl.reshape(cl.shape[0], -1)
    return cl[np.lexsort(flat[:, ::-1].T)]


def write_window(tmp, poses, frames):
    """the shipped formats: alidarPose.csv = 4 lines per pose, rows of [R|t] with a trailing comma, element
    (3,3) = timestamp; full<m>.pcd = 11 header lines + 32-byte records x y z intensity nx ny nz curvature"""
    R, p = npo.pose_R(poses), npo.pose_p(poses)
    with open(os.path.join(tmp, "alidarPose.csv"), "w") as f:
        for m in range(poses.shape[0]):
            for r in range(3):
                f.write("%.9f,%.9f,%.9f,%.9f,\n" % (R[m, r, 0], R[m, r, 1], R[m, r, 2], p[m, r]))
            f.write("0.000000,0.000000,0.000000,%.6f,\n" % (1630577758.5 + 0.5 * m))
    for m, xyz in enumerate(frames):
        n = xyz.shape[0]
        rec = np.zeros((n, 8), dtype=np.float32)
        rec[:, :3] = xyz
        rec[:, 3] = m
        hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity normal_x normal_y "
               "normal_z curvature\nSIZE 4 4 4 4 4 4 4 4\nTYPE F F F F F F F F\nCOUNT 1 1 1 1 1 1 1 1\nWIDTH %d\n"
               "HEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (n, n))
        with open(os.path.join(tmp, "full%d.pcd" % m), "wb") as f:
            f.write(hdr.encode())
            f.write(rec.tobytes())


def synthetic_window(seed, W, F, pts):
    sc = scene.generate(seed, W, F, pts, surf_range=15.0, keep_points=True)
    # odometry-grade start: a fifth of the generator's pose noise
    R0, Ri = npo.pose_R(sc.poses_gt), npo.pose_R(sc.poses_init)
    p0, pi = npo.pose_p(sc.poses_gt), npo.pose_p(sc.poses_init)
    R = np.stack([R0[i] @ npo.exp_so3(0.2 * _log(R0[i].T @ Ri[i])) for i in range(W)])
    poses = npo.make_poses(R, p0 + 0.2 * (pi - p0))
    frames = [sc.points[:, i].reshape(-1, 3).copy() for i in range(W)]
    return poses, frames


def _log(R):
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    k = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return 0.5 * k if th < 1e-3 else 0.5 * th / np.sin(th) * k


def test_readers_roundtrip(tmp_path):
    poses, frames = synthetic_window(3, 6, 20, 30)
    write_window(str(tmp_path), poses, frames)
    got_poses, stamps = rw.read_pose_csv(str(tmp_path / "alidarPose.csv"))
    assert got_poses.shape == (6, 12) and np.allclose(got_poses, poses, atol=1e-8)
    assert np.allclose(np.diff(stamps), 0.5)
    for m in range(6):
        assert np.array_equal(rw.read_pcd_xyz(str(tmp_path / ("full%d.pcd" % m))), frames[m])
    rel = rw.relative_to_first(got_poses)
    assert np.allclose(npo.pose_R(rel)[0], np.eye(3), atol=1e-8) and np.allclose(npo.pose_p(rel)[0], 0)


@needs_ref
@pytest.mark.parametrize("seed,W,F,pts,voxel", [(1, 12, 80, 40, 1.0), (2, 20, 150, 30, 2.0), (5, 8, 40, 60, 0.5)])
def test_association_matches_reference_on_synthetic_scans(tmp_path, seed, W, F, pts, voxel):
    poses, frames = synthetic_window(seed, W, F, pts)
    write_window(str(tmp_path), poses, frames)
    cl_r, fx_r, co_r, poses_r, npts = ref.realworld_features(str(tmp_path), voxel)
    poses_p, frames_p = rw.load_window(str(tmp_path))
    assert npts == sum(f.shape[0] for f in frames_p)
    assert np.abs(poses_p - poses_r).max() < 1e-14
    cl_p, co_p, layer = ah.associate(frames_p, poses_r, voxel)      # same poses bit for bit -> same voxel keys
    assert cl_p.shape == cl_r.shape and cl_p.shape[0] > 0
    assert np.array_equal(canon(cl_p), canon(cl_r))                 # bit-exact feature set
    assert np.array_equal(np.sort(co_p), np.sort(co_r))
    assert not (fx_r[:, 9] > 0).any()


def test_association_matches_reference_on_shipped_data():
    """the shipped benchmark_realworld window (177 scans) against the fixture made by the reference's code"""
    from conftest import ROOT
    data = os.environ.get("BALM_REFERENCE_ROOT", "/root/reference") + "/datas/benchmark_realworld"
    fix = os.path.join(ROOT, "oracle", "_ref", "realworld_features.npz")
    if not (os.path.isdir(data) and os.path.exists(fix)):
        pytest.skip("shipped data or oracle/_ref/realworld_features.npz not present")
    g = dict(np.load(fix))
    poses, frames = rw.load_window(data)
    assert np.abs(poses - g["poses"]).max() < 1e-13
    cl, co, layer = ah.associate(frames, g["poses"], 2.0)
    assert cl.shape == g["clusters"].shape == (2281, 177, 10)
    assert np.array_equal(canon(cl), canon(g["clusters"]))
    assert list(np.bincount(layer)) == [797, 449, 1035]


# ---- the consistency driver's association rules (N4): src/simulation/BAs_left.hpp copy of the state machine ----
def exact_plane_scans(seed, n_scans, n_planes, pts):
    """simulator-like scans: points exactly on planes (the strict test wants lambda0 < 1e-10, max distance < 1 mm)"""
    rng = np.random.default_rng(seed)
    R = np.stack([npo.exp_so3(0.03 * rng.standard_normal(3)) for _ in range(n_scans)])
    p = np.cumsum(0.1 * rng.standard_normal((n_scans, 3)), axis=0)
    R[0], p[0] = np.eye(3), 0
    normals = rng.standard_normal((n_planes, 3))
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    centers = rng.uniform(-8, 8, (n_planes, 3))
    frames = []
    for i in range(n_scans):
        w = []
        for k in range(n_planes):
            a = np.cross(normals[k], [0.3, 0.5, 0.8]); a /= np.linalg.norm(a)
            b = np.cross(normals[k], a)
            uv = rng.uniform(-0.35, 0.35, (pts, 2))
            w.append(centers[k] + uv[:, :1] * a + uv[:, 1:] * b)
        w.append(rng.uniform(-8, 8, (200, 3)))                       # clutter: never a plane
        w = np.concatenate(w)
        frames.append(((w - p[i]) @ R[i]).astype(np.float32))
    return npo.make_poses(R, p), frames


def _canon_with_fix(cl, fix):
    both = np.concatenate([cl.reshape(cl.shape[0], -1), fix], axis=1)
    return both[np.lexsort(both[:, ::-1].T)]


def test_consistency_rules_match_reference_association(tmp_path):
    from oracle import ref_sim
    if not ref_sim.available():
        pytest.skip("oracle/_ref/libbalm_ref_sim.so not built (needs /root/reference)")
    poses, frames = exact_plane_scans(4, 9, 40, 60)
    cl, co, layer, fix, pts = ah.associate(frames, poses, want_points=True, **rw.SIM_RULES)
    clr, fxr = ref_sim.associate(frames, poses, 1, 1.0)
    assert cl.shape == clr.shape and cl.shape[0] >= 10 and cl.shape[1] == 8
    assert np.array_equal(_canon_with_fix(cl, fix), _canon_with_fix(clr, fxr))      # bit-exact feature set, fix included
    assert (fix[:, 9] > 0).any() and not layer.any()                                # layer_limit 0: root voxels only
    # the exported points rebuild the exported clusters
    xyz, fid, sid = pts
    N = np.zeros(cl.shape[:2])
    np.add.at(N, (fid, sid), 1)
    assert np.array_equal(N, cl[..., 9])
    a, i = fid[0], sid[0]
    sel = (fid == a) & (sid == i)
    assert np.allclose(xyz[sel].astype(np.float64).sum(0), cl[a, i, 6:9], rtol=1e-12)


def test_consistency_rules_on_shipped_scans():
    from conftest import ROOT
    from oracle import ref_sim
    path = os.path.join(ROOT, "oracle", "_ref", "consistency_scans.npz")
    if not (os.path.exists(path) and ref_sim.available()):
        pytest.skip("oracle/_ref/consistency_scans.npz or libbalm_ref_sim.so not built")
    d = np.load(path)
    frames = np.split(d["xyz"], np.cumsum(d["counts"])[:-1])
    cl, co, layer, fix, _ = ah.associate(frames, d["poses"], **rw.SIM_RULES)
    clr, fxr = ref_sim.associate(frames, d["poses"], 1, 1.0)
    assert cl.shape == clr.shape == (1096, 100, 10)
    assert np.array_equal(_canon_with_fix(cl, fix), _canon_with_fix(clr, fxr))


# ---- committed golden vectors from both compiled copies of the reference's state machine (make_golden_assoc.py) ----
def _golden(name):
    from conftest import ROOT
    d = dict(np.load(os.path.join(ROOT, "tests", "golden", name)))
    d["frames"] = np.split(d["xyz"], np.cumsum(d["counts"])[:-1])
    return d


def test_host_association_matches_golden_benchmark_rules():
    g = _golden("assoc_bench_w8.npz")
    cl, co, layer = ah.associate(g["frames"], g["poses"], 1.0)
    assert cl.shape == g["clusters"].shape and np.array_equal(canon(cl), canon(g["clusters"]))
    assert np.array_equal(np.sort(co), np.sort(g["coeffs"])) and len(set(layer.tolist())) == 3


def test_host_association_matches_golden_consistency_rules():
    g = _golden("assoc_sim_w8.npz")
    cl, co, layer, fix, _ = ah.associate(g["frames"], g["poses"], **rw.SIM_RULES)
    assert cl.shape == g["clusters"].shape
    assert np.array_equal(_canon_with_fix(cl, fix), _canon_with_fix(g["clusters"], g["fix"]))


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")
def test_reference_octree_used_incrementally():
    """oracle/ref_driver.cpp's ref_win_* (the comparator of balm_window_*, tests/test_gpu_window.py): the reference's own
    cut_voxel / recut / marginalize / tras_opt driven scan by scan.  (i) one recut per scan is NOT the batch association
    in general (a voxel cut on the evidence of few scans stays cut), but it must be deterministic and, on a window whose
    every voxel is judged the same way from the first scan on, give the batch result; (ii) a marginalisation moves the
    first scans of plane voxels into world-frame fix clusters and the scans down."""
    from test_gpu_voxel import cluttered_window
    poses, frames = exact_plane_scans(4, 8, 40, 60)
    W = len(frames)
    win = ref.Window(W, voxel_size=1.0, layer_limit=0)
    for i in range(W):
        win.add_scan(frames[i], poses[i])
    cl, fix, co = win.features()
    win.close()
    cl_h, co_h, lay_h = ah.associate(frames, poses, 1.0, (1.0 / 16, 1.0 / 16, 1.0 / 9), 0, 15)
    assert cl.shape == cl_h.shape and cl.shape[0] > 5
    assert np.array_equal(canon(cl), canon(cl_h)) and not fix.any()
    # sliding: two windows of a cluttered scene
    poses, frames = cluttered_window(3, 10, 40, 120, 1500)
    runs = []
    for _ in range(2):
        win = ref.Window(8, voxel_size=1.0)
        for i in range(8):
            win.add_scan(frames[i], poses[i])
        n0 = win.features()[0].shape[0]
        win.marginalize(2, poses[:8])
        for i in range(8, 10):
            win.add_scan(frames[i], poses[i])
        cl, fix, co = win.features()
        win.close()
        runs.append((n0, cl, fix, co))
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
    n0, cl, fix, co = runs[0]
    assert cl.shape[1] == 8 and (fix[:, 9] > 0).sum() > 50
    assert np.array_equal(co, cl[..., 9].sum(1))                  # push_voxel's weight: the window's points (bavoxel.hpp:42-44)
    assert (cl[:, 6:, 9].sum(0) > 0).all()                        # the two new scans sit in the last slots
