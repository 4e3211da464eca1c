"""N3: adaptive-voxel association on the GPU (balm_associate, csrc/kernels_voxel.hip) against
the host association (oracle/host_association.cpp, itself pinned bit-exact to the reference's cut_voxel / recut /
tras_opt in test_association.py) and against a fixture made by the reference's own code from the shipped
scans.  Index / integer work: the feature SET is compared bit for bit (every cluster, every N); the order
differs by design (hash-map order vs key order)."""
import os

import numpy as np
import pytest

from balm_amd import capi
from balm_amd import realworld as rw
from oracle import assoc_host as ah
from conftest import ROOT
from oracle import numpy_oracle as npo
from test_association import canon, synthetic_window

pytestmark = pytest.mark.gpu


def cluttered_window(seed, W, n_planes, pts_per_plane, n_clutter):
    """scans of a box-like scene: planar patches of several sizes (so that root voxels, octants and
    sub-octants all become features) plus uniform clutter (non-planar voxels that split and die)"""
    rng = np.random.default_rng(seed)
    R = np.stack([npo.exp_so3(0.02 * rng.standard_normal(3)) for _ in range(W)])
    p = np.cumsum(0.15 * rng.standard_normal((W, 3)), axis=0)
    R[0], p[0] = np.eye(3), 0
    frames = []
    normals = rng.standard_normal((n_planes, 3))
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    centers = rng.uniform(-12, 12, (n_planes, 3))
    sizes = rng.choice([0.3, 0.8, 2.5], n_planes)
    for i in range(W):
        world = []
        for k in range(n_planes):
            a = np.cross(normals[k], [0.3, 0.5, 0.8]); a /= np.linalg.norm(a)
            b = np.cross(normals[k], a)
            uv = rng.uniform(-sizes[k], sizes[k], (pts_per_plane, 2))
            world.append(centers[k] + uv[:, :1] * a + uv[:, 1:] * b + 0.01 * rng.standard_normal((pts_per_plane, 1)) * normals[k])
        world.append(rng.uniform(-12, 12, (n_clutter, 3)))
        world = np.concatenate(world)
        world = world[rng.permutation(world.shape[0])]
        frames.append(((world - p[i]) @ R[i]).astype(np.float32))      # body = R^T (w - p)
    return npo.make_poses(R, p), frames


def check_same_features(ctx, frames, poses, voxel, thr=(1.0 / 16, 1.0 / 16, 1.0 / 9), min_ps=15):
    cl_h, co_h, layer_h = ah.associate(frames, poses, voxel, thr, 2, min_ps)
    F, nroots, feats = rw.associate_gpu(ctx, frames, poses, voxel, thr, min_ps=min_ps)
    assert F == cl_h.shape[0]
    if F == 0:
        return cl_h, layer_h
    cl_g, co_g, layer_g = feats
    assert np.array_equal(canon(cl_g), canon(cl_h))                    # bit-exact feature set
    assert np.array_equal(np.sort(co_g), np.sort(co_h))
    assert np.array_equal(np.bincount(layer_g, minlength=3), np.bincount(layer_h, minlength=3))
    assert np.array_equal(co_g, cl_g[..., 9].sum(1))
    return cl_g, layer_g


# (voxel sizes that are powers of two take the exact-multiply form of cut_voxel's key, the others its division: voxel_key)
@pytest.mark.parametrize("seed,W,voxel", [(1, 8, 1.0), (2, 20, 2.0), (3, 5, 4.0), (4, 33, 0.5), (5, 9, 0.7), (6, 12, 1.3)])
def test_device_association_matches_host_on_cluttered_scans(seed, W, voxel):
    poses, frames = cluttered_window(seed, W, 60, 120, 3000)
    c = capi.Context(W)
    cl, layer = check_same_features(c, frames, poses, voxel)
    assert cl.shape[0] > 10
    if voxel <= 2.0:
        assert len(set(layer.tolist())) >= 2                            # several octree depths exercised
    c.close()


def test_device_association_feeds_the_optimizer():
    """installing through balm_associate == installing the host association's features through balm_set_features"""
    poses, frames = synthetic_window(1, 20, 150, 40)
    c = capi.Context(20)
    cl, _ = check_same_features(c, frames, poses, 1.0)
    out_g, lg_g = c.damping_iter(poses, form=0, u0=0.01, max_iter=10, min_planes=20)
    cl_h, co_h, _ = ah.associate(frames, poses, 1.0)
    c2 = capi.Context(20)
    c2.set_features(cl_h, None, co_h)
    out_h, lg_h = c2.damping_iter(poses, form=0, u0=0.01, max_iter=10, min_planes=20)
    assert len(lg_g) == len(lg_h)
    assert np.abs(lg_g[:, :2] - lg_h[:, :2]).max() <= 1e-9 * lg_h[0, 0]     # feature order changes the summation order only
    assert np.abs(out_g - out_h).max() < 1e-6          # well inside the path's pose tolerance (1e-5 rad / 1e-4 m)
    c.close(); c2.close()


def test_association_table_is_fetched_lazily_and_survives_a_new_table():
    """round 3: balm_associate leaves the feature table on the device; balm_get_features fetches the clusters on first use --
    also after balm_set_features has replaced the table the optimizer works on (the host copy is taken before d_cl is
    overwritten), and twice in a row"""
    import ctypes as C
    poses, frames = synthetic_window(3, 12, 120, 30)
    c = capi.Context(12)
    xyz = np.concatenate(frames).astype(np.float32)
    fid = np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])
    F, nroot, feats = c.associate(xyz, fid, poses, voxel_size=1.0)              # fetches at once: the reference copy
    cl_ref, co_ref, lay_ref = feats
    F2, _, _ = c.associate(xyz, fid, poses, voxel_size=1.0, want_features=False)   # nothing fetched
    assert F2 == F
    other = np.zeros((5, 12, 10)); other[:, :, 9] = 3.0; other[:, :, 0] = other[:, :, 3] = other[:, :, 5] = 1.0
    other[:, :, 6] = np.arange(12)[None, :] * 0.1
    c.set_features(other, None, np.ones(5))                                         # d_cl now holds another table
    for _ in range(2):
        cl, co, lay = np.zeros_like(cl_ref), np.zeros_like(co_ref), np.zeros_like(lay_ref)
        c._check(c.L.balm_get_features(c.h, cl.ctypes.data_as(C.c_void_p), co.ctypes.data_as(C.c_void_p), lay.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(cl, cl_ref) and np.array_equal(co, co_ref) and np.array_equal(lay, lay_ref)
    c.close()


def test_device_association_edge_cases():
    c = capi.Context(4)
    # nothing planar / too few points: zero features is a result, not an error; the context stays usable
    rng = np.random.default_rng(0)
    frames = [rng.uniform(-3, 3, (10, 3)).astype(np.float32) for _ in range(4)]
    poses = npo.make_poses(np.stack([np.eye(3)] * 4), np.zeros((4, 3)))
    F, nroots, feats = rw.associate_gpu(c, frames, poses, 1.0)
    assert F == 0 and feats is None and nroots > 0
    with pytest.raises(capi.BalmError):
        c.evaluate(0, poses)
    # a plane seen by ONE scan only is not a feature (push_voxel needs two observers, bavoxel.hpp:32-37);
    # negative coordinates exercise the `loc < 0 -> loc - 1` key rule; empty scans are legal
    uv = rng.uniform(-0.4, 0.4, (200, 2))
    plane = np.column_stack([uv[:, 0] - 5.5, uv[:, 1] - 7.5, np.full(200, -2.5) + 0.002 * rng.standard_normal(200)])
    frames = [plane.astype(np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)]
    assert rw.associate_gpu(c, frames, poses, 1.0)[0] == 0 == ah.associate(frames, poses, 1.0)[0].shape[0]
    frames[2] = plane[::-1].astype(np.float32).copy()
    check_same_features(c, frames, poses, 1.0)
    assert c.F >= 1
    with pytest.raises(capi.BalmError):                                    # frame id out of range
        c.associate(plane.astype(np.float32), np.full(200, 4, np.int32), poses, 1.0)
    bad = plane.astype(np.float32).copy()
    bad[17, 1] = np.nan
    with pytest.raises(capi.BalmError):                                    # non-finite point: rejected, not hashed
        c.associate(bad, np.zeros(200, np.int32), poses, 1.0)
    check_same_features(c, frames, poses, 1.0)                             # the context survives both
    c.close()


def test_device_association_matches_reference_on_shipped_scans():
    """first 24 scans of the shipped benchmark_realworld window (1.76 M points) against the features the
    reference's own cut_voxel/recut/tras_opt produced from them (tools/make_realworld_fixture.py)"""
    path = os.path.join(ROOT, "oracle", "_ref", "realworld_scans_w24.npz")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/realworld_scans_w24.npz not built (needs /root/reference/datas)")
    g = np.load(path)
    counts = g["counts"]
    frames = np.split(g["xyz"], np.cumsum(counts)[:-1])
    c = capi.Context(len(counts), flags=capi.FLAG_TIMING)
    F, nroots, (cl, co, layer) = rw.associate_gpu(c, frames, g["poses"], 2.0)
    assert cl.shape == g["clusters"].shape
    assert np.array_equal(canon(cl), canon(g["clusters"]))
    assert np.array_equal(np.sort(co), np.sort(g["coeffs"]))
    ms, n = c.timing()["voxel"]
    print("balm_associate: %d points -> %d root voxels, %d features, device %.2f ms" % (counts.sum(), nroots, F, ms))
    c.close()


def test_benchmark_realworld_end_to_end_from_raw_scans():
    """BASELINE configs[4] with nothing on the CPU between the files and the result: the 177 shipped scans
    (13.4 M points) -> balm_associate -> device LM loop, against the reference's feature set (bit for bit)
    and the poses the reference's BALM2::damping_iter reaches from them."""
    scans = os.path.join(ROOT, "oracle", "_ref", "realworld_scans_w177.npz")
    feats = os.path.join(ROOT, "oracle", "_ref", "realworld_features.npz")
    if not (os.path.exists(scans) and os.path.exists(feats)):
        pytest.skip("oracle/_ref/realworld_scans_w177.npz not built (needs /root/reference/datas)")
    from util import ROT_TOL_RAD, TRANS_TOL_M, pose_errors
    g, sdat = np.load(feats), np.load(scans)
    counts = sdat["counts"]
    xyz = sdat["xyz"]
    fid = np.repeat(np.arange(len(counts), dtype=np.int32), counts)
    c = capi.Context(len(counts), flags=capi.FLAG_TIMING)
    c.associate(xyz, fid, g["poses"], 2.0, want_features=False)          # warm-up: sizes the scratch arena
    c.reset_timing()
    F, nroots, (cl, co, layer) = c.associate(xyz, fid, g["poses"], 2.0)
    ms = c.timing()["voxel"][0]
    assert cl.shape == g["clusters"].shape == (2281, 177, 10)
    assert np.array_equal(canon(cl), canon(g["clusters"]))
    assert list(np.bincount(layer)) == [797, 449, 1035]
    out, lg = c.damping_iter(g["poses"], form=0, u0=0.01, max_iter=10, min_planes=20)
    assert len(lg) == len(g["ref_log"])
    rot, tr = pose_errors(out, g["ref_poses"])
    assert rot.max() <= ROT_TOL_RAD and tr.max() <= TRANS_TOL_M
    print("realworld from raw scans: %d points -> %d roots -> %d features in %.2f ms (reference CPU %.1f s); "
          "LM %d iterations; pose diff %.1e rad %.1e m"
          % (xyz.shape[0], nroots, F, ms, float(g["ref_seconds_association"]), len(lg), rot.max(), tr.max()))
    c.close()


def test_device_association_wide_keys():
    """fine voxels over a large extent: the packed root key needs more than 32 bits and there are more than 2^17
    root voxels, so both sorts take their 64-bit-key path (the common case runs on 32-bit keys)"""
    rng = np.random.default_rng(11)
    W = 3
    poses = npo.make_poses(np.stack([npo.exp_so3(0.001 * rng.standard_normal(3)) for _ in range(W)]),
                           0.0005 * rng.standard_normal((W, 3)))
    frames = []
    for i in range(W):
        uv = rng.uniform(-0.003, 0.003, (400, 2))
        patch = np.column_stack([5.002 + uv[:, 0], -3.004 + uv[:, 1], np.full(400, 1.005) + 2e-5 * rng.standard_normal(400)])
        clutter = rng.uniform(-12, 12, (70000, 3))
        pts = np.concatenate([patch, clutter])[rng.permutation(70400)]
        R, p = npo.pose_R(poses)[i], npo.pose_p(poses)[i]
        frames.append(((pts - p) @ R).astype(np.float32))
    c = capi.Context(W)
    cl, layer = check_same_features(c, frames, poses, 0.01)
    assert cl.shape[0] >= 1
    c.close()


def _canon_with_fix(cl, fix):
    both = np.concatenate([cl.reshape(cl.shape[0], -1), fix], axis=1)
    return both[np.lexsort(both[:, ::-1].T)]


@pytest.mark.parametrize("layer_limit,min_observers,fix_frames", [(0, 2, 0), (1, 1, 0), (2, 0, 2), (1, 2, 1)])
def test_device_association_rule_options(layer_limit, min_observers, fix_frames):
    """the association's globals as options: layer_limit (bavoxel.hpp:8), push_voxel's observer minimum (:32-37),
    marginalisation of the first scans into world-frame fix clusters (to_margi :778-816, batch form) -- against the
    host association with the same rules; and the point -> feature map against the clusters it must rebuild"""
    W = 10
    poses, frames = cluttered_window(9, W + fix_frames, 50, 100, 2500)
    kw = dict(voxel_size=1.0, layer_limit=layer_limit, min_observers=min_observers, fix_frames=fix_frames)
    cl_h, co_h, lay_h, fix_h, _ = ah.associate(frames, poses, want_points=True, **kw)
    c = capi.Context(W)
    F, nroots, (cl, co, layer, fix, pf) = rw.associate_gpu(c, frames, poses, want_points=True, **kw)
    assert F == cl_h.shape[0] and F > 5 and cl.shape == (F, W, 10)
    assert np.array_equal(_canon_with_fix(cl, fix), _canon_with_fix(cl_h, fix_h))
    assert layer.max() <= layer_limit and np.array_equal(np.bincount(layer, minlength=3), np.bincount(lay_h, minlength=3))
    assert (fix[:, 9] > 0).any() == (fix_frames > 0)
    # point -> feature map: counting the mapped points per (feature, scan) gives back every N
    scan = np.concatenate([np.full(f.shape[0], i) for i, f in enumerate(frames)]) - fix_frames
    keep = (pf >= 0) & (scan >= 0)
    N = np.zeros((F, W))
    np.add.at(N, (pf[keep], scan[keep]), 1)
    assert np.array_equal(N, cl[..., 9])
    # the features (fix clusters included) are installed: optimiser runs
    out, lg = c.damping_iter(poses[fix_frames:], form=0, u0=0.01, max_iter=3)
    assert np.isfinite(out).all()
    c.close()


def test_device_association_consistency_rules():
    """the consistency driver's rule set (strict plane test on exact planes, layer_limit 0, first scan marginalised,
    no observer minimum) against the host association, which tests/test_association.py pins to the reference's
    compiled copy"""
    from test_association import exact_plane_scans
    poses, frames = exact_plane_scans(4, 9, 40, 60)
    cl_h, co_h, lay_h, fix_h, _ = ah.associate(frames, poses, **rw.SIM_RULES)
    c = capi.Context(8)
    F, nroots, (cl, co, layer, fix, pf) = rw.associate_gpu(c, frames, poses, want_points=True, **rw.SIM_RULES)
    assert F == cl_h.shape[0] >= 10
    assert np.array_equal(_canon_with_fix(cl, fix), _canon_with_fix(cl_h, fix_h))
    # without the strict test the clutter-contaminated voxels would pass the eigen-ratio test too
    loose = dict(rw.SIM_RULES, strict=None)
    assert rw.associate_gpu(c, frames, poses, **loose)[0] >= F
    c.close()


def test_device_association_matches_reference_golden_vectors():
    """committed golden vectors made by both compiled copies of the reference's association (tests/golden/
    make_golden_assoc.py): benchmark rules (all three layers) and the consistency driver's rules (fix clusters)"""
    from test_association import _golden
    g = _golden("assoc_bench_w8.npz")
    c = capi.Context(8)
    F, _, (cl, co, layer) = rw.associate_gpu(c, g["frames"], g["poses"], 1.0)
    assert cl.shape == g["clusters"].shape and np.array_equal(canon(cl), canon(g["clusters"]))
    assert np.array_equal(np.sort(co), np.sort(g["coeffs"]))
    g = _golden("assoc_sim_w8.npz")
    F, _, (cl, co, layer, fix, _) = rw.associate_gpu(c, g["frames"], g["poses"], **rw.SIM_RULES)
    assert np.array_equal(_canon_with_fix(cl, fix), _canon_with_fix(g["clusters"], g["fix"]))
    c.close()


def test_device_association_points_not_in_scan_order():
    """the level sorts drop the scan bits only when the points arrive scan by scan; a shuffled input must take the full
    sorts and still give the same FEATURE SET (sums inside a cluster follow the input order of its points, so values
    agree to rounding, counts and layers exactly)"""
    poses, frames = cluttered_window(5, 9, 40, 120, 400)
    c = capi.Context(9)
    F0, _, (cl0, co0, lay0) = rw.associate_gpu(c, frames, poses, 1.0)
    xyz = np.concatenate(frames)
    fid = np.concatenate([np.full(f.shape[0], i, dtype=np.int32) for i, f in enumerate(frames)])
    perm = np.random.default_rng(1).permutation(xyz.shape[0])
    F1, _, (cl1, co1, lay1) = c.associate(xyz[perm], fid[perm], poses, 1.0)
    assert F0 == F1 and np.array_equal(lay0, lay1) and np.array_equal(co0, co1)
    assert np.array_equal(cl0[..., 9], cl1[..., 9])
    assert np.abs(cl0 - cl1).max() <= 1e-12 * np.abs(cl0).max()
    c.close()


@pytest.mark.parametrize("seed,W,voxel,layer_limit,opts", [
    (1, 8, 1.0, 2, {}), (2, 20, 2.0, 2, {}), (6, 12, 1.0, 1, {}), (7, 6, 2.0, 0, {}),
    (8, 10, 1.0, 2, dict(fix_frames=2, min_observers=0, want_points=True)),
    (9, 7, 1.0, 2, dict(strict=(0.05, 25.0, 1e-2), want_points=True)),
])
def test_partition_path_is_the_sorted_path_bit_for_bit(seed, W, voxel, layer_limit, opts, monkeypatch):
    """Round 5's association (records laid down in root order, levels 1-2 as stable partitions inside the root voxels, segment sums
    streaming the records) against the library-sort path it replaces (BALM_ASSOC=sorted: composite keys + rocPRIM radix sorts per
    level + gathers): the SAME feature table in the same order, bit for bit -- clusters, weights, layers, fix clusters, and the
    feature of every point -- with every rule option on the way (layer limits, marginalised scans, the strict plane test)."""
    poses, frames = cluttered_window(seed, W, 50, 150, 2500)
    ff = opts.get("fix_frames", 0)

    def run():
        c = capi.Context(W - ff)
        out = rw.associate_gpu(c, frames, poses, voxel, layer_limit=layer_limit, **opts)
        c.close()
        return out

    F_a, nr_a, feats_a = run()
    # BALM_ASSOC=radix: round 5's root order (radix sort of (key, index) pairs + gather) instead of round 6's stable multisplit
    for other in ("sorted", "radix"):
        monkeypatch.setenv("BALM_ASSOC", other)
        F_b, nr_b, feats_b = run()
        assert F_a == F_b and nr_a == nr_b and F_a > (0 if opts.get("strict") else 5)
        assert len(feats_a) == len(feats_b)
        for x, y in zip(feats_a, feats_b):
            assert (x is None and y is None) or np.array_equal(x, y)


@pytest.mark.gpu
def test_multisplit_root_order_on_ragged_tiles_and_many_roots(monkeypatch):
    """the multisplit's corners: a last tile of a few points, one-point scans, a window with more root voxels than its LDS tables hold
    (falls back to the radix order) -- always the table of BALM_ASSOC=radix, bit for bit, with the feature of every point"""
    def both(frames, poses, voxel, W):
        outs = []
        for mode in ("", "radix"):
            if mode:
                monkeypatch.setenv("BALM_ASSOC", mode)
            else:
                monkeypatch.delenv("BALM_ASSOC", raising=False)
            c = capi.Context(W)
            outs.append(rw.associate_gpu(c, frames, poses, voxel, want_points=True))
            c.close()
        (Fa, na, fa), (Fb, nb, fb) = outs
        assert Fa == Fb and na == nb
        for x, y in zip(fa or (), fb or ()):
            assert (x is None and y is None) or np.array_equal(x, y)
        return Fa, na
    poses, frames = cluttered_window(31, 7, 40, 150, 2500)
    frames[3] = frames[3][:1]                                   # a one-point scan
    frames[6] = frames[6][:8192 * 2 + 3 - sum(len(f) for f in frames[:6]) % 8192] if len(frames[6]) > 20000 else frames[6]
    F, nroots = both(frames, poses, 4.0, 7)
    assert F > 2 and nroots <= 2048
    # a fine grid: thousands of roots -> the radix path by itself
    F2, nroots2 = both(frames, poses, 1.0, 7)
    assert F2 > 5 and nroots2 > 2048


@pytest.mark.gpu
@pytest.mark.parametrize("assoc", ["", "sorted"])
def test_head_scan_that_gives_up_waiting_counts_for_itself(assoc, monkeypatch):
    """k_scan_heads (the one-pass scan of the per-point key lists) takes a workgroup's INDEX as its tile and bounds the wait for its
    predecessors' words; a workgroup that gives up counts the heads in front of its tile itself.  BALM_SCAN_SPIN=1 makes every workgroup
    give up at its first look: the same feature table, bit for bit (several tiles per list: 150 000+ points, and a ragged last tile)."""
    poses, frames = cluttered_window(21, 12, 50, 150, 14000)
    if assoc:
        monkeypatch.setenv("BALM_ASSOC", assoc)

    def run():
        c = capi.Context(12)
        out = rw.associate_gpu(c, frames, poses, 1.0, layer_limit=2, want_points=True)
        c.close()
        return out

    F_a, nr_a, feats_a = run()
    monkeypatch.setenv("BALM_SCAN_SPIN", "1")
    F_b, nr_b, feats_b = run()
    assert sum(len(f) for f in frames) > 8 * 8192 and F_a == F_b and nr_a == nr_b and F_a > 5
    for x, y in zip(feats_a, feats_b):
        assert (x is None and y is None) or np.array_equal(x, y)

