"""Sliding-window map (balm_window_*, csrc/kernels_window.inc): the reference's octree used INCREMENTALLY -- cut_voxel into
a live map, one recut per scan, tras_opt, marginalize with re-transformed poses, repeat -- against the reference's own
OCTO_TREE_ROOT compiled from bavoxel.hpp (oracle/_ref, ref_driver.cpp ref_win_*), call for call.  The per-scan clusters of
the features are sums of points in scan order: compared bit for bit as a set.  Fix clusters go through
PointCluster::transform (a different rounding path than sums of points, and not exactly symmetric): 1e-12 relative."""
import numpy as np
import pytest

from balm_amd import capi
from oracle import orc, ref
from oracle import numpy_oracle as npo
from test_gpu_voxel import cluttered_window

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")]


def canon_order(cl):
    flat = cl.reshape(cl.shape[0], -1)
    return np.lexsort(flat[:, ::-1].T)


def compare(ctx, win, tag):
    F, feats = ctx.window_features()
    cl_r, fix_r, co_r = win.features()
    assert F == cl_r.shape[0], "%s: %d features, reference %d" % (tag, F, cl_r.shape[0])
    if F == 0:
        return 0, 0
    cl, co, layer, fix = feats
    og, orf = canon_order(cl), canon_order(cl_r)
    assert np.array_equal(cl[og], cl_r[orf]), tag                       # bit-exact per-scan clusters, same feature set
    assert np.array_equal(co[og], co_r[orf]), tag
    scale = np.abs(fix_r[orf]).max(axis=1, keepdims=True) + 1e-300
    assert np.all(np.abs(fix[og] - fix_r[orf]) <= 1e-12 * scale), tag
    assert np.array_equal(fix[og][:, 9], fix_r[orf][:, 9]), tag
    return F, int((fix[:, 9] > 0).sum())


def noisy(poses, rng, s_rot, s_tr):
    """poses [k,12] (column-major R | p) with a small left perturbation: stands in for what an optimiser hands back"""
    out = poses.copy()
    for i in range(poses.shape[0]):
        R = poses[i, :9].reshape(3, 3).T
        Rn = npo.exp_so3(s_rot * rng.standard_normal(3)) @ R
        out[i, :9] = Rn.T.reshape(9)
        out[i, 9:] = poses[i, 9:] + s_tr * rng.standard_normal(3)
    return out


@pytest.mark.parametrize("seed,W,mg,retransform", [(3, 8, 2, True), (5, 6, 1, True), (7, 8, 3, False)])
def test_sliding_window_matches_reference_octree(seed, W, mg, retransform):
    slides = 4
    total = W + slides * mg
    poses, frames = cluttered_window(seed, total, 40, 120, 1500)
    rng = np.random.default_rng(seed)
    start = noisy(poses, rng, 2e-3, 2e-2)                 # odometry-grade initial poses: the map is cut with these
    ctx = capi.Context(W)
    ctx.window_open(voxel_size=1.0)
    win = ref.Window(W, voxel_size=1.0)
    cur = []                                               # poses of the scans in the window, as the map knows them
    for i in range(W):
        ctx.window_add_scan(frames[i], start[i]); win.add_scan(frames[i], start[i]); cur.append(start[i])
    F0, _ = compare(ctx, win, "full window")
    assert F0 > 10
    nxt = W
    saw_fix = 0
    for sl in range(slides):
        # "optimised" poses: closer to the truth than the odometry ones
        xs = None
        if retransform:
            xs = noisy(poses[nxt - W:nxt], rng, 2e-4, 2e-3)
            cur = list(xs)
        ctx.window_marginalize(mg, xs); win.marginalize(mg, xs)
        cur = cur[mg:]
        assert ctx.window_info()[0] == W - mg
        for k in range(mg):
            ctx.window_add_scan(frames[nxt], start[nxt]); win.add_scan(frames[nxt], start[nxt]); cur.append(start[nxt])
            nxt += 1
        F, nfix = compare(ctx, win, "slide %d" % sl)
        assert F > 10
        saw_fix = max(saw_fix, nfix)
    assert saw_fix > 0                                     # marginalised scans live on as fix clusters
    # the installed table (fix clusters included) feeds the optimiser
    out, lg = ctx.damping_iter(np.stack(cur), form=0, u0=0.01, max_iter=3)
    assert np.isfinite(out).all()
    win.close(); ctx.window_close(); ctx.close()


@pytest.mark.parametrize("seed,W,mg", [(11, 8, 2), (13, 6, 3)])
def test_staged_recut_equals_the_level_by_level_recut(seed, W, mg, monkeypatch):
    """round 3: balm_window_add_scan's recut runs as stages (all levels of the unseen scan at once, then the children of
    fresh cuts from the list k_win_descend leaves, one host synchronisation per stage); the level-by-level form of round 2
    stays selectable (BALM_WINDOW_RECUT=levels) as its comparator: same features, clusters bit for bit, same fix clusters,
    same point -> feature map, after every slide (only the node ids differ)"""
    slides = 3
    total = W + slides * mg
    poses, frames = cluttered_window(seed, total, 40, 120, 1500)
    rng = np.random.default_rng(seed)
    start = noisy(poses, rng, 2e-3, 2e-2)
    xs_all = [noisy(poses[sl * mg:W + sl * mg], rng, 2e-4, 2e-3) for sl in range(slides)]

    def run(mode):
        if mode:
            monkeypatch.setenv("BALM_WINDOW_RECUT", mode)
        else:
            monkeypatch.delenv("BALM_WINDOW_RECUT", raising=False)
        ctx = capi.Context(W)
        ctx.window_open(voxel_size=1.0)
        out = []

        def snap():
            F, feats = ctx.window_features()
            cl, co, layer, fix = feats
            o = canon_order(cl)
            out.append((F, cl[o], co[o], layer[o], fix[o], ctx.window_info()[1:]))
        for i in range(W):
            ctx.window_add_scan(frames[i], start[i])
        snap()
        nxt = W
        for sl in range(slides):
            ctx.window_marginalize(mg, xs_all[sl])
            for k in range(mg):
                ctx.window_add_scan(frames[nxt], start[nxt]); nxt += 1
            snap()
        ctx.window_close(); ctx.close()
        return out

    a, b = run("levels"), run(None)
    assert len(a) == len(b)
    for (Fa, cla, coa, la, fa, ia), (Fb, clb, cob, lb, fb, ib) in zip(a, b):
        assert Fa == Fb and Fa > 10 and ia == ib
        assert np.array_equal(cla, clb) and np.array_equal(coa, cob) and np.array_equal(la, lb) and np.array_equal(fa, fb)


def test_window_first_fill_equals_batch_when_nothing_is_cut_early():
    """adding the scans one by one with a recut each is NOT the batch association in general (a voxel cut on the evidence of
    three scans stays cut) -- but on exact, well separated planes no voxel changes its mind, and both must agree"""
    from test_association import exact_plane_scans
    from balm_amd import realworld as rw
    poses, frames = exact_plane_scans(4, 8, 40, 60)
    W = len(frames)
    ctx = capi.Context(W)
    ctx.window_open(voxel_size=1.0, layer_limit=0, min_observers=2)
    for i in range(W):
        ctx.window_add_scan(frames[i], poses[i])
    F, (cl, co, layer, fix) = ctx.window_features()
    Fb, _, (clb, cob, layb) = rw.associate_gpu(ctx, frames, poses, voxel_size=1.0, layer_limit=0)
    assert F == Fb and F > 5
    assert np.array_equal(cl[canon_order(cl)], clb[canon_order(clb)])
    assert not fix.any()
    ctx.close()


def test_window_argument_errors():
    ctx = capi.Context(4)
    with pytest.raises(capi.BalmError):
        ctx.window_add_scan(np.zeros((10, 3), np.float32), np.zeros(12))        # no open window
    ctx.window_open(voxel_size=1.0)
    pose = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=float)
    pts = np.random.default_rng(0).uniform(-3, 3, (200, 3)).astype(np.float32)
    for _ in range(4):
        ctx.window_add_scan(pts, pose)
    with pytest.raises(capi.BalmError):
        ctx.window_add_scan(pts, pose)                                         # window full
    with pytest.raises(capi.BalmError):
        ctx.window_marginalize(5)
    bad = pts.copy(); bad[3, 1] = np.nan
    ctx.window_marginalize(1)
    with pytest.raises(capi.BalmError):
        ctx.window_add_scan(bad, pose)
    assert ctx.window_info()[0] == 3
    ctx.close()


def test_sliding_window_on_shipped_scans():
    """the shipped benchmark_realworld scans as a stream: a 20-scan window sliding by 5 over the first 45 scans (76 k points
    per scan, voxel 2 m), poses re-estimated at every slide -- against the reference's octree, call for call"""
    import os
    import time
    from conftest import ROOT
    scans = os.path.join(ROOT, "oracle", "_ref", "realworld_scans_w177.npz")
    feats = os.path.join(ROOT, "oracle", "_ref", "realworld_features.npz")
    if not (os.path.exists(scans) and os.path.exists(feats)):
        pytest.skip("oracle/_ref/realworld_scans_w177.npz not built (needs /root/reference/datas)")
    sdat, g = np.load(scans), np.load(feats)
    counts = sdat["counts"]
    frames = np.split(sdat["xyz"], np.cumsum(counts)[:-1])
    poses = g["poses"]
    W, mg, total = 20, 5, 45
    rng = np.random.default_rng(1)
    ctx = capi.Context(W, flags=capi.FLAG_TIMING)
    ctx.window_open(voxel_size=2.0)
    win = ref.Window(W, voxel_size=2.0)
    t_gpu = t_ref = 0.0
    def add(i):
        nonlocal t_gpu, t_ref
        t0 = time.perf_counter(); ctx.window_add_scan(frames[i], poses[i]); t1 = time.perf_counter()
        win.add_scan(frames[i], poses[i]); t2 = time.perf_counter()
        t_gpu += t1 - t0; t_ref += t2 - t1
    for i in range(W):
        add(i)
    F, nfix = compare(ctx, win, "first window")
    layers = []
    nxt = W
    while nxt + mg <= total:
        xs = noisy(poses[nxt - W:nxt], rng, 1e-4, 1e-3)
        ctx.window_marginalize(mg, xs); win.marginalize(mg, xs)
        for k in range(mg):
            add(nxt); nxt += 1
        F, nfix = compare(ctx, win, "window ending at scan %d" % nxt)
        layers.append((F, nfix))
    assert F > 300 and nfix > 100
    _, (cl, co, layer, fix) = ctx.window_features()
    assert len(np.unique(layer)) == 3                      # features at all three octree layers
    scans_in, pts, nodes = ctx.window_info()
    print("sliding window on shipped scans: %d scans, %d points and %d nodes resident; features/fix per slide %s; layers %s; "
          "add_scan (cut_voxel + recut) %.1f ms per scan on the device call, reference %.1f ms"
          % (scans_in, pts, nodes, layers, list(np.bincount(layer)), 1e3 * t_gpu / total, 1e3 * t_ref / total))
    win.close(); ctx.close()


@pytest.mark.parametrize("W,slide,total", [(20, 5, 40), (30, 10, 120)])
def test_sliding_window_ba_on_shipped_scans_against_the_reference_loop(W, slide, total):
    """BASELINE configs[4]'s sliding window end to end: balm_amd.sliding.SlidingWindowBA (map, features and LM loop on the
    device) over the first 40 (W = 20 sliding by 5) and 120 (W = 30 sliding by 10) shipped scans -- against the same loop on the CPU: the reference's own
    octree (cut_voxel / recut / tras_opt / marginalize, compiled from bavoxel.hpp) and the oracle's LM loop on every window.
    Why not bavoxel.hpp's BALM2::damping_iter itself: its left_evaluate_acc2 starts C from zero (:325) while its
    evaluate_only_residual starts from the fix cluster (:443) -- with non-empty fix clusters r1 and r2 of one iteration
    measure different things (the first slide: gain ratio 15.7 / 2.5).  benchmark_virtual.cpp:241-243 and BAs_left.hpp carry
    the consistent evaluator (fix cluster in both), which is what the oracle restates and the device computes.  The oracle
    re-anchors its result to pose 0 (:1159-1164), which a window with world-frame fix clusters must not do; the comparison
    applies that re-anchoring to the device's poses instead."""
    import os
    from conftest import ROOT
    from balm_amd.sliding import SlidingWindowBA, compose, inverse
    from util import ROT_TOL_RAD, TRANS_TOL_M, pose_errors
    scans = os.path.join(ROOT, "oracle", "_ref", "realworld_scans_w177.npz")
    feats = os.path.join(ROOT, "oracle", "_ref", "realworld_features.npz")
    if not (os.path.exists(scans) and os.path.exists(feats)):
        pytest.skip("oracle/_ref/realworld_scans_w177.npz not built (needs /root/reference/datas)")
    sdat, g = np.load(scans), np.load(feats)
    counts = sdat["counts"]
    frames = np.split(sdat["xyz"], np.cumsum(counts)[:-1])
    odom = g["poses"]
    ctx = capi.Context(W)
    ba = SlidingWindowBA(ctx, slide, voxel_size=2.0)
    win = ref.Window(W, voxel_size=2.0)
    worst = [0.0, 0.0]
    nwin = 0
    for i in range(total):
        guess_before = None
        r = ba.push(frames[i], odom[i])
        # the reference's map sees the same scan with the same pose guess (what push() handed to balm_window_add_scan)
        guess_before = (r["poses_in"][-1] if r is not None else ba.est[-1])
        win.add_scan(frames[i], guess_before)
        if r is None:
            continue
        nwin += 1
        cl_r, fix_r, co_r = win.features()
        assert cl_r.shape[0] == r["F"]
        out_r, lg_r = orc.damping_iter(0, cl_r, fix_r, co_r, r["poses_in"], 0.01, 10)
        assert len(lg_r) == len(r["log"]) and np.array_equal(lg_r[:, 6], r["log"][:, 6])
        assert np.allclose(lg_r[:, :2], r["log"][:, :2], rtol=1e-8)
        anchored = np.stack([compose(inverse(r["poses"][0]), p) for p in r["poses"]])
        rot, tr = pose_errors(anchored, out_r)
        worst = [max(worst[0], rot.max()), max(worst[1], tr.max())]
        assert rot.max() <= ROT_TOL_RAD and tr.max() <= TRANS_TOL_M
        win.marginalize(slide, r["poses"])
    assert nwin == (total - W) // slide + 1 and ba.trajectory().shape == (total, 12)
    print("sliding-window BA, %d windows of %d scans: device vs reference loop %.1e rad %.1e m" % (nwin, W, worst[0], worst[1]))
    win.close(); ctx.close()


def test_window_features_feed_a_sharded_context():
    """balm_window_features on a multi-device context (here: three loopback shards on the one GPU) hands the window's
    feature table -- fix clusters included -- to the shards; the LM run must land where the single context lands"""
    from util import pose_errors
    W, mg = 8, 2
    poses, frames = cluttered_window(11, W + mg, 40, 120, 1500)
    outs = []
    for kw in (dict(), dict(flags=capi.FLAG_LOOPBACK_SHARDS, n_devices=3)):
        ctx = capi.Context(W, 0, **kw) if kw else capi.Context(W)
        ctx.window_open(voxel_size=1.0)
        for i in range(W):
            ctx.window_add_scan(frames[i], poses[i])
        ctx.window_marginalize(mg, poses[:W])
        for i in range(W, W + mg):
            ctx.window_add_scan(frames[i], poses[i])
        F, (cl, co, layer, fix) = ctx.window_features()
        assert F > 10 and (fix[:, 9] > 0).any()
        out, lg = ctx.damping_iter(poses[mg:], form=0, u0=0.01, max_iter=5, reanchor=False)
        outs.append((F, out, lg))
        ctx.close()
    assert outs[0][0] == outs[1][0] and len(outs[0][2]) == len(outs[1][2])
    rot, tr = pose_errors(outs[0][1], outs[1][1])
    assert rot.max() < 1e-10 and tr.max() < 1e-10


def test_window_edge_cases_against_reference():
    """small and degenerate sequences, call for call against the reference's octree: a tiny scan, a scan that only revisits
    known voxels, layer_limit 0 and 1, the whole window marginalised at once and the map used again afterwards"""
    poses, frames = cluttered_window(21, 9, 30, 80, 600)
    for limit in (0, 1):
        ctx = capi.Context(4)
        ctx.window_open(voxel_size=1.0, layer_limit=limit)
        win = ref.Window(4, voxel_size=1.0, layer_limit=limit)
        def both(f, *a):
            getattr(ctx, "window_" + f)(*a); getattr(win, f)(*a)
        both("add_scan", frames[0], poses[0])
        both("add_scan", frames[1][:10], poses[1])                # ten points
        both("add_scan", frames[0], poses[0])                     # the first scan again: only known voxels
        both("add_scan", frames[2], poses[2])
        compare(ctx, win, "limit %d, first window" % limit)
        both("marginalize", 4, np.stack([poses[0], poses[1], poses[0], poses[2]]))        # everything leaves
        assert ctx.window_info()[:2] == (0, 0)
        compare(ctx, win, "limit %d, empty window" % limit)      # no window points: no features on either side
        for i in range(3, 7):
            both("add_scan", frames[i], poses[i])
        F, nfix = compare(ctx, win, "limit %d, refilled" % limit)
        assert F > 5 and nfix > 0                                  # the fix clusters of the first life are still there
        both("marginalize", 1, None)
        both("add_scan", frames[7], poses[7])
        compare(ctx, win, "limit %d, after a marginalisation without poses" % limit)
        win.close(); ctx.close()


def _canon_fix(cl, fix):
    both = np.concatenate([cl.reshape(cl.shape[0], -1), fix], axis=1)
    return both[np.lexsort(both[:, ::-1].T)]


@pytest.mark.parametrize("seed,W,fix", [(4, 8, 1), (9, 6, 2)])
def test_window_map_with_the_consistency_drivers_rules(seed, W, fix):
    """The one driver of the reference that really uses its octree incrementally (src/simulation/consistency.cpp:108-136):
    cut_voxel for win_size + fix_size scans, ONE recut, ONE marginalize(fix_size, {}, win_count) -- with that file's copy
    of the map (BAs_left.hpp: the strict plane test :647-674 over the points' distances, fix_point.N < 30 :756, a cut_voxel
    that keeps every point :1120) -- on the device map (fix_frames, strict, fix_point_limit, defer_recut) against the
    compiled BAs_left.hpp octree call for call, against the batch balm_associate with the same rules, and onward: more
    scans, one recut each, another marginalisation."""
    from oracle import ref_sim
    from balm_amd import realworld as rw
    from test_association import exact_plane_scans
    if not ref_sim.available():
        pytest.skip("oracle/_ref/libbalm_ref_sim.so not built")
    extra = 2 * fix
    poses, frames = exact_plane_scans(seed, W + fix + extra, 40, 60)
    R = rw.SIM_RULES
    ctx = capi.Context(W)
    ctx.window_open(voxel_size=R["voxel_size"], eigen_thresholds=R["eigen_thresholds"], min_ps=R["min_ps"], layer_limit=R["layer_limit"],
                    min_observers=R["min_observers"], fix_frames=fix, strict=R["strict"], fix_point_limit=30, defer_recut=True)
    win = ref_sim.Window(W, fix, R["voxel_size"])
    for i in range(W + fix):
        ctx.window_add_scan(frames[i], poses[i])
        win.cut_voxel(frames[i], poses[i])
    with pytest.raises(capi.BalmError):                    # scans no recut has seen, and more than `win` of them
        ctx.window_features()
    ctx.window_recut(); win.recut()
    ctx.window_marginalize(fix); win.marginalize(fix)
    F, (cl, co, layer, fx) = ctx.window_features()
    cl_r, fx_r = win.features()
    assert F == cl_r.shape[0] >= 10 and (fx[:, 9] > 0).any()
    assert np.array_equal(_canon_fix(cl, fx), _canon_fix(cl_r, fx_r))          # no re-transform: fix clusters are sums too -> bit for bit
    # the batch form with the same rules (pinned to the reference's compiled association in tests/test_association.py)
    c2 = capi.Context(W)
    rules = dict(R, fix_frames=fix)
    Fb, _, (clb, cob, layb, fxb, _) = rw.associate_gpu(c2, frames[:W + fix], poses[:W + fix], want_points=True, **rules)
    assert Fb == F and np.array_equal(_canon_fix(clb, fxb), _canon_fix(cl, fx))
    c2.close()
    # onward: the map stays alive -- scan, recut, scan, recut, marginalise (the calling convention of bavoxel.hpp on this map)
    for rnd in range(2):
        for k in range(fix):
            i = W + fix + rnd * fix + k
            ctx.window_add_scan(frames[i], poses[i]); ctx.window_recut()
            win.cut_voxel(frames[i], poses[i]); win.recut()
        ctx.window_marginalize(fix); win.marginalize(fix)
        F, (cl, co, layer, fx) = ctx.window_features()
        cl_r, fx_r = win.features()
        assert F == cl_r.shape[0] and np.array_equal(_canon_fix(cl, fx), _canon_fix(cl_r, fx_r)), rnd
    win.close(); ctx.close()
