"""GPU parity tests proper: every call goes through the C ABI (include/balm_hip.h) into the HIP
kernels and is compared with the CPU oracle on the same seeded inputs.

Tolerances (FP64 on both sides; only the summation order differs, SURVEY.md 7 hard-part 6):
  Hessian / gradient / residual : 1e-10 relative to the largest entry
  LM trace (r1, r2, u)          : 1e-8 relative
  final poses                   : BASELINE.json north_star: rotation <= 1e-5 rad, translation <= 1e-4 m
"""
import os

import numpy as np
import pytest

from balm_amd import capi, scene
from oracle import orc
from util import ROT_TOL_RAD, TRANS_TOL_M, make_scene, pose_errors, rel_err

pytestmark = pytest.mark.gpu

HTOL = 1e-10


def ctx_for(sc, fix=None, flags=0):
    c = capi.Context(sc.W, 0, flags)
    c.set_features(sc.clusters, fix, sc.coeffs)
    return c


CASES = [
    # (seed, W, F, pts, drop, with_fix)
    (1, 20, 20, 40, 0.0, False),      # BASELINE configs[0]: launch default
    (2, 7, 9, 12, 0.3, True),         # ragged window (n=42 < one tile), sparse, fix clusters
    (3, 33, 50, 8, 0.5, False),       # W not a multiple of anything, half the observations gone
    (4, 64, 300, 6, 0.0, False),      # configs[1] shape, fewer features
    (5, 100, 120, 6, 0.2, True),      # several tiles, padded last tile (600 -> 640)
]


@pytest.mark.parametrize("form", [0, 1])
@pytest.mark.parametrize("case", CASES)
def test_evaluate_matches_oracle(case, form):
    seed, W, F, pts, drop, wf = case
    sc, fix = make_scene(seed, W, F, pts, drop, wf)
    c = ctx_for(sc, fix)
    H, g, r = c.evaluate(form, sc.poses_init)
    Ho, go, ro = orc.evaluate(form, sc.clusters, fix, sc.coeffs, sc.poses_init)
    assert abs(r - ro) / ro < 1e-12
    assert rel_err(g, go) < HTOL
    assert rel_err(H, Ho) < HTOL
    if form == 0:
        assert np.array_equal(H, H.T)                      # mirrored exactly (bavoxel.hpp:422-424)
    # residual-only kernel path
    r2 = c.only_residual(sc.poses_init)
    assert abs(r2 - orc.only_residual(sc.clusters, fix, sc.coeffs, sc.poses_init)) / ro < 1e-12
    c.close()


@pytest.mark.parametrize("W,F,form", [(177, 60, 0), (230, 40, 0), (230, 40, 1), (320, 30, 0), (480, 24, 0), (480, 24, 1),
                                      (500, 16, 0), (700, 10, 1), (1024, 6, 0)])
def test_wide_windows(W, F, form):
    """windows beyond one LDS default (W > 210), the shipped data's W=177, W=480 (the last window whose per-pose
    accumulators fit one workgroup's LDS) and pose-chunked windows up to the context limit of 1024 poses;
    evaluate + one damped solve against the oracle"""
    sc, _ = make_scene(90 + W, W, F, 4, drop=0.3, mode=1)
    c = ctx_for(sc)
    H, g, r = c.evaluate(form, sc.poses_init)
    Ho, go, ro = orc.evaluate_threads(form, sc.clusters, None, sc.coeffs, sc.poses_init, 8)
    assert abs(r - ro) / ro < 1e-12 and rel_err(g, go) < HTOL and rel_err(H, Ho) < HTOL
    dx, q1 = c.solve_damped(Ho, go, 0.5)
    A = Ho + 0.5 * np.diag(np.diag(Ho))
    assert np.linalg.norm(A @ dx + go) / np.linalg.norm(go) < 1e-8
    c.close()


def test_evaluate_is_deterministic_run_to_run():
    sc, _ = make_scene(11, 40, 200, 6)
    c = ctx_for(sc)
    H1, g1, r1 = c.evaluate(0, sc.poses_init)
    H2, g2, r2 = c.evaluate(0, sc.poses_init)
    assert np.array_equal(H1, H2) and np.array_equal(g1, g2) and r1 == r2
    c.close()


def test_feature_subranges_add_up():
    """the reference splits [0,F) over threads and sums (bavoxel.hpp:1044-1056)"""
    sc, _ = make_scene(12, 20, 61, 10, drop=0.2)
    c = ctx_for(sc)
    H, g, r = c.evaluate(0, sc.poses_init)
    part = 1.0 * sc.F / 4
    Hs, gs, rs = 0, 0, 0
    for t in range(4):
        Ht, gt, rt = c.evaluate(0, sc.poses_init, int(part * t), int(part * (t + 1)))
        Ho, go, ro = orc.evaluate(0, sc.clusters, None, sc.coeffs, sc.poses_init, int(part * t), int(part * (t + 1)))
        assert rel_err(Ht, Ho) < HTOL and abs(rt - ro) / ro < 1e-12
        Hs, gs, rs = Hs + Ht, gs + gt, rs + rt
    assert rel_err(Hs, H) < 1e-12 and rel_err(gs, g) < 1e-12 and abs(rs - r) / r < 1e-13
    c.close()


@pytest.mark.parametrize("W,F,drop", [(20, 37, 0.2), (177, 900, 0.6), (200, 3000, 0.0)])
def test_set_features_through_the_fill_callback_is_the_flat_upload(W, F, drop):
    """balm_set_features_cb (include/balm_shim.hpp's upload: VOX_HESS's borrowed per-feature vectors pulled straight into the
    library's pinned chunks by its host threads) installs the same table as balm_set_features on the flat array: H, g, r bit for
    bit, same work model and sparse plan; the 3 000-feature table (48 MB) takes the multi-chunk ring path."""
    sc, _ = make_scene(77, W, F, 6, drop=drop)
    a = capi.Context(W, 0, capi.FLAG_TIMING)
    a.set_features(sc.clusters, None, sc.coeffs)
    Ha, ga, ra = a.evaluate(0, sc.poses_init)
    b = capi.Context(W, 0, capi.FLAG_TIMING)
    b.set_features_cb([sc.clusters[k].copy() for k in range(F)], None, sc.coeffs)
    Hb, gb, rb = b.evaluate(0, sc.poses_init)
    assert np.array_equal(Ha, Hb) and np.array_equal(ga, gb) and ra == rb
    assert a.work_model() == b.work_model()
    assert a.timing()["upload"][1] == 1 and b.timing()["upload"][1] == 1 and a.timing()["upload"][0] > 0
    Ho, go, ro = orc.evaluate_threads(0, sc.clusters, None, sc.coeffs, sc.poses_init, 8)
    assert rel_err(Hb, Ho) < HTOL and rel_err(gb, go) < HTOL and abs(rb - ro) / ro < 1e-12
    # a second table through the same ring (its chunks may still be draining), smaller and larger than the first
    for F2 in (max(2, F // 3), F):
        b.set_features_cb([sc.clusters[k].copy() for k in range(F2)], None, sc.coeffs[:F2])
        a.set_features(sc.clusters[:F2], None, sc.coeffs[:F2])
        assert np.array_equal(a.evaluate(0, sc.poses_init)[0], b.evaluate(0, sc.poses_init)[0])
    a.close(); b.close()


def test_linearity_in_weights():
    """size-independent property: H, g, r are linear in the feature weights"""
    sc, _ = make_scene(13, 24, 80, 6)
    c = ctx_for(sc)
    H1, g1, r1 = c.evaluate(0, sc.poses_init)
    c.set_features(sc.clusters, None, 3.0 * sc.coeffs)
    H3, g3, r3 = c.evaluate(0, sc.poses_init)
    assert rel_err(H3, 3 * H1) < 1e-12 and rel_err(g3, 3 * g1) < 1e-12 and abs(r3 - 3 * r1) / r1 < 1e-12
    c.close()


@pytest.mark.parametrize("W,u", [(20, 0.1), (20, 0.01), (33, 1e-4), (100, 0.01)])
def test_solve_damped_matches_oracle(W, u):
    sc, _ = make_scene(20 + W, W, 3 * W, 6)
    Ho, go, _ = orc.evaluate(0, sc.clusters, None, sc.coeffs, sc.poses_init)
    A = Ho + u * np.diag(np.diag(Ho))
    c = capi.Context(W)
    dx, q1 = c.solve_damped(Ho, go, u)
    dxo, q1o = orc.solve_damped(Ho, go, u)
    # the noisy start is indefinite (SURVEY.md finding 4): both sides must *solve* it
    assert np.linalg.norm(A @ dx + go) / np.linalg.norm(go) < 1e-9
    assert rel_err(dx, dxo) < 1e-7
    assert abs(q1 - q1o) / abs(q1o) < 1e-8
    c.close()


def test_solve_indefinite_system():
    sc, _ = make_scene(31, 12, 10, 10, drop=0.3)
    Ho, go, _ = orc.evaluate(0, sc.clusters, None, sc.coeffs, sc.poses_init)
    u = 0.01
    A = Ho + u * np.diag(np.diag(Ho))
    assert np.linalg.eigvalsh(A).min() < 0
    c = capi.Context(sc.W)
    dx, _ = c.solve_damped(Ho, go, u)
    assert np.linalg.norm(A @ dx + go) / np.linalg.norm(go) < 1e-8
    dxo, _ = orc.solve_damped(Ho, go, u)
    assert rel_err(dx, dxo) < 1e-6
    c.close()


LM_CASES = [
    # (seed, W, F, pts, drop, form, u0, max_iter)
    (1, 20, 20, 40, 0.0, 0, 0.1, 20),     # benchmark_virtual constants
    (1, 20, 20, 40, 0.0, 0, 0.01, 10),    # bavoxel constants (indefinite first step)
    (1, 20, 20, 40, 0.0, 1, 0.1, 20),     # right form
    (6, 30, 150, 10, 0.4, 0, 0.01, 10),   # sparse co-visibility
    (7, 64, 400, 6, 0.0, 0, 0.1, 20),     # configs[1] shape
    (8, 64, 400, 6, 0.0, 1, 0.1, 20),     # right form at configs[1]'s window (acc_evaluate2 + the right update, bavoxel.hpp:1119-1120)
    (9, 96, 300, 6, 0.3, 1, 0.01, 10),    # right form, sparse co-visibility, bavoxel constants
]


@pytest.mark.parametrize("case", LM_CASES)
def test_damping_iter_follows_oracle_trajectory(case):
    seed, W, F, pts, drop, form, u0, mi = case
    sc, _ = make_scene(seed, W, F, pts, drop)
    c = ctx_for(sc)
    out, lg = c.damping_iter(sc.poses_init, form=form, u0=u0, max_iter=mi)
    oo, lo = orc.damping_iter(form, sc.clusters, None, sc.coeffs, sc.poses_init, u0, mi)
    assert len(lg) == len(lo)
    assert np.array_equal(lg[:, 6], lo[:, 6])                    # same accept / reject decisions
    assert np.allclose(lg[:, 0], lo[:, 0], rtol=1e-8)            # r1
    assert np.allclose(lg[:, 1], lo[:, 1], rtol=1e-8)            # r2
    assert np.allclose(lg[:, 2], lo[:, 2], rtol=1e-6)            # u
    rot, tr = pose_errors(out, oo)
    assert rot.max() <= ROT_TOL_RAD and tr.max() <= TRANS_TOL_M
    # and both land on the ground truth (reference's printed metric, benchmark_virtual.cpp:517-518)
    gt = orc.reanchor(sc.poses_gt)
    r_g, t_g = orc.rsme(gt, out)
    r_o, t_o = orc.rsme(gt, oo)
    assert abs(r_g - r_o) < 1e-6 and abs(t_g - t_o) < 1e-6
    c.close()


@pytest.mark.parametrize("case", [
    # (seed, W, F, pts, drop, with_fix, form, u0, max_iter, graph)
    (1, 20, 20, 40, 0.0, False, 0, 0.01, 10, False),
    (1, 20, 20, 40, 0.0, False, 0, 0.01, 10, True),     # replayed hipGraphs hold the (swapped) factor buffers
    (2, 7, 30, 12, 0.3, True, 0, 0.1, 12, False),       # fix clusters, ragged window
    (6, 30, 150, 10, 0.4, False, 1, 0.01, 10, False),   # right form, sparse plan (slot order of the columns)
    (7, 64, 400, 6, 0.0, False, 0, 0.1, 20, False),
    (11, 200, 900, 6, 0.5, False, 0, 0.1, 8, False),    # the bench window's width, half the observations gone
    (12, 256, 300, 6, 0.2, True, 1, 0.1, 6, False),     # the widest window the one-pass kernel takes
    (13, 300, 200, 6, 0.2, False, 0, 0.1, 6, False),    # wider: both runs take K1 + K1b + K2 (the switch is a no-op)
])
def test_one_pass_trial_evaluation_matches_the_three_kernel_path(case, monkeypatch):
    """k_moments_factors (round 3: the trial evaluation of the LM loop reads the clusters ONCE and leaves residual, eigen
    records AND the factors G~ / diagonal partials of the trial poses, which an accepted step's Hessian evaluation starts
    from) against the K1 + K1b -> K2 sequence (the default; the one-pass kernel is BALM_FUSE_TRIAL=1): same decisions, residuals to rounding,
    same poses; and against the oracle.  Then the context must still evaluate correctly (no stale factors are reused)."""
    seed, W, F, pts, drop, wf, form, u0, mi, graph = case
    sc, fix = make_scene(seed, W, F, pts, drop, wf)
    if graph:
        monkeypatch.setenv("BALM_GRAPH", "1")
    c = ctx_for(sc, fix)
    monkeypatch.delenv("BALM_FUSE_TRIAL", raising=False)
    pa, la = c.damping_iter(sc.poses_init, form=form, u0=u0, max_iter=mi, min_planes=0)
    monkeypatch.setenv("BALM_FUSE_TRIAL", "1")          # opt-in (it is the slower of the two on the box, see balm_capi.hip)
    pb, lb = c.damping_iter(sc.poses_init, form=form, u0=u0, max_iter=mi, min_planes=0)
    monkeypatch.delenv("BALM_FUSE_TRIAL")               # and back: graphs and buffers of the other mode must not leak in
    pc, lc = c.damping_iter(sc.poses_init, form=form, u0=u0, max_iter=mi, min_planes=0)
    assert np.array_equal(pa, pc) and np.array_equal(la, lc)
    assert len(la) == len(lb) and np.array_equal(la[:, 6], lb[:, 6])
    assert np.allclose(la[:, :2], lb[:, :2], rtol=1e-10, atol=0) and np.allclose(la[:, 2], lb[:, 2], rtol=1e-7)
    assert np.abs(pa - pb).max() < 1e-9
    assert la[:, 6].sum() >= 2                          # accepted steps: the factor buffers were swapped in
    oo, lo = orc.damping_iter(form, sc.clusters, fix, sc.coeffs, sc.poses_init, u0, mi)
    assert len(lb) == len(lo) and np.allclose(lb[:, :2], lo[:, :2], rtol=1e-8)
    rot, tr = pose_errors(pb, oo)
    assert rot.max() <= ROT_TOL_RAD and tr.max() <= TRANS_TOL_M
    H, g, r = c.evaluate(form, sc.poses_init)
    Ho, go, ro = orc.evaluate(form, sc.clusters, fix, sc.coeffs, sc.poses_init)
    assert rel_err(H, Ho) < HTOL and rel_err(g, go) < HTOL and abs(r - ro) / ro < 1e-12
    c.close()


def test_too_few_planes_is_an_error_code_not_exit():
    sc, _ = make_scene(40, 10, 12, 6)
    c = ctx_for(sc)
    with pytest.raises(capi.BalmError) as e:
        c.damping_iter(sc.poses_init, min_planes=20)              # bavoxel.hpp:1079-1085
    assert e.value.code == capi.ERR_TOO_FEW_PLANES
    c.close()


def test_call_order_and_argument_errors():
    c = capi.Context(8)
    with pytest.raises(capi.BalmError) as e:
        c.evaluate(0, np.zeros((8, 12)))
    assert e.value.code == capi.ERR_STATE
    sc, _ = make_scene(41, 8, 5, 6)
    c.set_features(sc.clusters, None, sc.coeffs)
    with pytest.raises(capi.BalmError) as e:
        c.evaluate(0, sc.poses_init, 3, 2)
    assert e.value.code == capi.ERR_ARG
    with pytest.raises(capi.BalmError):
        c.evaluate(2, sc.poses_init)
    c.close()


def test_build_clusters_matches_push():
    """N1: GPU cluster build == PointCluster::push over the same points (tools.hpp:311-316)"""
    sc = scene.generate(50, 12, 30, 9, keep_points=True)
    F, W, pts = sc.F, sc.W, sc.pts
    xyz = sc.points.reshape(-1, 3)
    fid = np.repeat(np.arange(F, dtype=np.int32), W * pts)
    pid = np.tile(np.repeat(np.arange(W, dtype=np.int32), pts), F)
    c = capi.Context(W)
    got = c.build_clusters(F, xyz, fid, pid, None, sc.coeffs)
    # grouped points: every cluster is pushed point by point in order by one lane with the reference's operation
    # sequence -> bit-identical to the host's PointCluster::push, every entry (and run to run)
    assert np.array_equal(got, sc.clusters)
    assert np.array_equal(c.build_clusters(F, xyz, fid, pid, None, sc.coeffs), got)
    # runs longer than a wavefront's 64 points and runs that straddle its windows
    sc2 = scene.generate(51, 5, 7, 150, keep_points=True)
    fid2 = np.repeat(np.arange(7, dtype=np.int32), 5 * 150)
    pid2 = np.tile(np.repeat(np.arange(5, dtype=np.int32), 150), 7)
    c2 = capi.Context(5)
    assert np.array_equal(c2.build_clusters(7, sc2.points.reshape(-1, 3), fid2, pid2, None, sc2.coeffs), sc2.clusters)
    c2.close()
    # shuffled input (runs broken up: the order-free build takes over) must give the same clusters
    perm = np.random.default_rng(0).permutation(xyz.shape[0])
    got2 = c.build_clusters(F, xyz[perm], fid[perm], pid[perm], None, sc.coeffs)
    assert np.array_equal(got2[..., 9], sc.clusters[..., 9])
    assert rel_err(got2, sc.clusters) < 1e-13
    # and the installed clusters drive the same evaluation
    _, g, r = c.evaluate(0, sc.poses_init, want_hess=False)
    _, go, ro = orc.evaluate(0, sc.clusters, None, sc.coeffs, sc.poses_init)
    assert abs(r - ro) / ro < 1e-12 and rel_err(g, go) < 1e-10
    c.close()


@pytest.mark.parametrize("terms", ["0", "1"])
@pytest.mark.parametrize("W,F,pts", [(6, 9, 40), (200, 30, 40), (5, 7, 150), (3, 4, 700), (2, 3, 2500), (7, 11, 6), (9, 5, 24)])
def test_build_clusters_long_runs_both_lane_mappings(W, F, pts, terms):
    """N1 with runs of 6 .. 2500 points per (feature, pose) -- the launch default is 40 (benchmark_virtual.launch:4-9) -- through
    both lane mappings of k_build_clusters_runs (one lane per run / one lane per (run, term column)) and through the
    continuation paths of a run that leaves its block (staged 64 points / chunked): bit-identical to PointCluster::push"""
    import os
    sc = scene.generate(60 + pts, W, F, pts, keep_points=True)
    xyz = sc.points.reshape(-1, 3)
    fid = np.repeat(np.arange(F, dtype=np.int32), W * pts)
    pid = np.tile(np.repeat(np.arange(W, dtype=np.int32), pts), F)
    c = capi.Context(W)
    os.environ["BALM_BUILD_TERMS"] = terms
    try:
        for bp in ("256", "512", ""):
            if bp:
                os.environ["BALM_BUILD_BP"] = bp
            else:
                os.environ.pop("BALM_BUILD_BP", None)
            got = c.build_clusters(F, xyz, fid, pid, None, sc.coeffs)
            assert np.array_equal(got, sc.clusters), (bp, np.abs(got - sc.clusters).max())
        # ragged runs: drop a prefix of every run's points (lengths 1 .. pts), keys stay grouped
        rng = np.random.default_rng(pts)
        keep = np.ones(xyz.shape[0], bool)
        for k in range(F * W):
            keep[k * pts: k * pts + int(rng.integers(0, pts))] = False
        got = c.build_clusters(F, xyz[keep], fid[keep], pid[keep], None, sc.coeffs)
        ref = np.zeros_like(sc.clusters)
        P = sc.points.reshape(F, W, pts, 3).astype(np.float64)
        K = keep.reshape(F, W, pts)
        for a in range(F):
            for i in range(W):
                acc = np.zeros(10)
                for q in P[a, i][K[a, i]]:       # the reference's push: one rounding per operation, in order
                    acc[:6] += np.array([q[0] * q[0], q[0] * q[1], q[0] * q[2], q[1] * q[1], q[1] * q[2], q[2] * q[2]])
                    acc[6:9] += q
                    acc[9] += 1
                ref[a, i] = acc
        assert np.array_equal(got, ref)
    finally:
        os.environ.pop("BALM_BUILD_TERMS", None); os.environ.pop("BALM_BUILD_BP", None)
    c.close()


def test_timing_slots_fill_when_enabled():
    sc, _ = make_scene(60, 20, 40, 6)
    c = ctx_for(sc, flags=capi.FLAG_TIMING)
    c.damping_iter(sc.poses_init, u0=0.1, max_iter=3, no_stop=True, force_hess=True)
    t = c.timing()
    assert t["syrk"][1] == 3 and t["syrk"][0] > 0
    # one residual evaluation per iteration + one moments pass per Hessian evaluation whose poses were
    # not just evaluated (after an accepted step the trial-pose records are reused)
    assert t["solve"][1] == 3 and 4 <= t["moments"][1] <= 6
    c.close()


@pytest.mark.parametrize("W,F", [(200, 600)])
def test_full_width_window_properties(W, F):
    """BASELINE configs[2] window (W=200, 15x15 tiles) at a feature count the oracle still
    finishes in seconds, plus size-independent properties: symmetry, gauge null-space of the
    exact Hessian at the optimum-free residual (translation gauge: H t = 0 does not hold for the
    left form in general, so we check the gradient's gauge orthogonality instead)."""
    sc, _ = make_scene(70, W, F, 6, mode=1)
    c = ctx_for(sc)
    H, g, r = c.evaluate(0, sc.poses_init)
    Ho, go, ro = orc.evaluate_threads(0, sc.clusters, None, sc.coeffs, sc.poses_init, 8)
    assert abs(r - ro) / ro < 1e-12 and rel_err(g, go) < HTOL and rel_err(H, Ho) < HTOL
    assert np.array_equal(H, H.T)
    # a common left-translation of every pose leaves the residual unchanged -> sum of the
    # translational gradient blocks vanishes
    gt = g.reshape(W, 6)[:, 3:].sum(0)
    assert np.abs(gt).max() < 1e-9 * np.abs(g).max()
    c.close()


def test_bench_size_properties():
    """BASELINE configs[2] at FULL size (W=200 poses, 50 000 features: the bench workload) through properties that
    do not need the oracle at that size: feature sub-ranges add up, symmetry, weight linearity, translation-gauge
    orthogonality of the gradient, the oracle on a random sample of the features through the sub-range entry, and
    an LM run that must reach the ground truth."""
    W, F = 200, 50000
    # (under BALM_SYRK=int8 -- the whole suite can be run with it -- the Hessian's tolerances are the switch's contract at this size, 1e-11 of the
    #  largest entry: a sub-range and a reweighted table slice to different digits; g and the residual do not pass through the product)
    htol_split, htol_lin = (1e-11, 1e-11) if os.environ.get("BALM_SYRK") == "int8" else (1e-12, 1e-13)
    sc = scene.generate(123, W, F, 6, mode=1)
    c = capi.Context(W)
    c.set_features(sc.clusters, None, sc.coeffs)
    H, g, r = c.evaluate(0, sc.poses_init)
    assert np.array_equal(H, H.T) and np.isfinite(H).all()
    # (1) sub-ranges add up (the reference's thread split, bavoxel.hpp:1044-1056)
    cut = 17321
    H1, g1, r1 = c.evaluate(0, sc.poses_init, 0, cut)
    H2, g2, r2 = c.evaluate(0, sc.poses_init, cut, F)
    assert rel_err(H1 + H2, H) < htol_split and rel_err(g1 + g2, g) < 1e-12 and abs(r1 + r2 - r) / r < 1e-13
    # (2) gauge: a common left translation leaves the residual unchanged
    assert np.abs(g.reshape(W, 6)[:, 3:].sum(0)).max() < 1e-9 * np.abs(g).max()
    # (3) the oracle on 48 consecutive features somewhere in the middle, via the sub-range entry
    lo = 31007
    Hs, gs, rs = c.evaluate(0, sc.poses_init, lo, lo + 48)
    Ho, go, ro = orc.evaluate_threads(0, sc.clusters[lo:lo + 48], None, sc.coeffs[lo:lo + 48], sc.poses_init, 8)
    assert abs(rs - ro) / ro < 1e-12 and rel_err(gs, go) < HTOL and rel_err(Hs, Ho) < HTOL
    # (4) residual-only == the evaluator's residual
    assert abs(c.only_residual(sc.poses_init) - r) / r < 1e-13
    # (5) weights enter linearly
    c2 = capi.Context(W)
    c2.set_features(sc.clusters, None, 3.0 * sc.coeffs)
    H3, g3, r3 = c2.evaluate(0, sc.poses_init)
    assert rel_err(H3, 3.0 * H) < htol_lin and abs(r3 - 3.0 * r) / r < 1e-13
    c2.close()
    # (6) the LM loop reaches the ground truth of the scene (RSME as benchmark_virtual.cpp:48-61 reports it)
    out, lg = c.damping_iter(sc.poses_init, form=0, u0=0.1, max_iter=20)
    assert lg[-1, 1] < 0.05 * lg[0, 0]
    rot, tr = orc.rsme(orc.reanchor(sc.poses_gt), out)
    print("W=200 F=50000: %d LM iterations, residual %.4g -> %.4g, RSME %.2e rad %.2e m" % (len(lg), lg[0, 0], lg[-1, 1], rot, tr))
    assert rot < 1e-4 and tr < 1e-3
    c.close()


# ---- golden fixtures: outputs of the reference's own source (tests/golden/make_golden.py) ------------
import glob as _glob
import os as _os

_GOLD = sorted(g for g in _glob.glob(_os.path.join(_os.path.dirname(__file__), "golden", "*.npz"))
               if not _os.path.basename(g).startswith(("cov_", "assoc_", "lm_big_", "window_")))      # covariance fixtures: tests/test_gpu_cov.py


@pytest.mark.parametrize("path", _GOLD, ids=[_os.path.basename(p)[:-4] for p in _GOLD])
def test_hip_matches_reference_fixtures(path):
    g = dict(np.load(path))
    cl, co, P = g["clusters"], g["coeffs"], g["poses"]
    c = capi.Context(cl.shape[1])
    c.set_features(cl, None, co)
    for form in (0, 1):
        H, J, r = c.evaluate(form, P)
        assert abs(r - g["r%d" % form]) / g["r%d" % form] < 1e-12
        assert rel_err(J, g["g%d" % form]) < HTOL
        if "H%d" % form in g:
            assert rel_err(H, g["H%d" % form]) < HTOL
    assert abs(c.only_residual(P) - g["r_only"]) / g["r_only"] < 1e-12
    for u in (0.01, 0.1):
        dx, q1 = c.solve_damped(g["H0"], g["g0"], u)
        assert rel_err(dx, g["dx_u%g" % u]) < 1e-7
        assert abs(q1 - g["q1_u%g" % u]) / abs(g["q1_u%g" % u]) < 1e-8
    if "lm_poses" in g:      # BALM2::damping_iter of the reference: left form, u0 = 0.01, <= 10 iterations
        out, lg = c.damping_iter(P, form=0, u0=0.01, max_iter=10, min_planes=20)
        assert len(lg) == len(g["lm_log"])
        assert np.allclose(lg[:, :2], g["lm_log"][:, :2], atol=2e-6)      # the reference prints 6 decimals
        rot, tr = pose_errors(out, g["lm_poses"])
        assert rot.max() <= ROT_TOL_RAD and tr.max() <= TRANS_TOL_M
    c.close()


def test_cpp_shim_dropin_with_reference_association():
    """include/balm_shim.hpp end to end: the reference's own cut_voxel/recut/tras_opt (compiled from
    /root/reference) fill VOX_HESS; the same container then runs through the reference BALM2 (CPU)
    and BALM2_HIP (GPU) -- tests/cpp/shim_driver.cpp.  Needs the binary built in the build container."""
    import subprocess
    from conftest import ROOT
    exe = _os.path.join(ROOT, "oracle", "_ref", "shim_driver")
    if not _os.path.exists(exe):
        pytest.skip("oracle/_ref/shim_driver not built (needs /root/reference at build time)")
    scene_so = _os.path.join(ROOT, "balm_amd", "lib", "libbalm_scene.so")
    p = subprocess.run([exe, "1", "20", "150", "40", scene_so], cwd=ROOT, capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("SHIM_DRIVER")]
    assert line, p.stdout[-2000:] + p.stderr[-2000:]
    print(line[-1])
    assert p.returncode == 0, line[-1]


def test_allreduce_hook_over_rccl_single_rank():
    """the multi-GPU plumbing with world_size 1: torch.distributed 'nccl' (= RCCL) all-reduce of the
    library's device payload through balm_set_allreduce must leave every result unchanged."""
    import socket
    import torch.distributed as dist
    from balm_amd import dist as bdist
    sc, _ = make_scene(80, 24, 90, 6, drop=0.2)
    c = ctx_for(sc)
    H0, g0, r0 = c.evaluate(0, sc.poses_init)
    out0, lg0 = c.damping_iter(sc.poses_init, u0=0.01, max_iter=10)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    _os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    bdist.init_process_group("nccl")
    try:
        hook = bdist.install_allreduce(c)
        H1, g1, r1 = c.evaluate(0, sc.poses_init)
        assert np.array_equal(H0, H1) and np.array_equal(g0, g1) and r0 == r1
        assert abs(c.only_residual(sc.poses_init) - r0) / r0 < 1e-14
        out1, lg1 = c.damping_iter(sc.poses_init, u0=0.01, max_iter=10)
        assert np.array_equal(out0, out1) and np.array_equal(lg0, lg1)
        print("zero-copy device view:", hook.zero_copy)
    finally:
        c.set_allreduce(None)
        dist.destroy_process_group()
    c.close()


def test_realworld_window_matches_reference_optimizer():
    """BASELINE configs[4] input: the shipped benchmark_realworld data (W=177 scans, 13.4 M points)
    associated by the reference's own cut_voxel/recut/tras_opt (tools/make_realworld_fixture.py ->
    oracle/_ref/realworld_features.npz: 2281 plane features, 15 % block fill).  The HIP LM loop must
    land on the poses of the reference's BALM2::damping_iter."""
    import time
    from conftest import ROOT
    path = _os.path.join(ROOT, "oracle", "_ref", "realworld_features.npz")
    if not _os.path.exists(path):
        pytest.skip("oracle/_ref/realworld_features.npz not built (needs /root/reference/datas)")
    g = dict(np.load(path))
    cl, co, P = g["clusters"], g["coeffs"], g["poses"]
    c = capi.Context(cl.shape[1])
    c.set_features(cl, None, co)
    out, lg = c.damping_iter(P, form=0, u0=0.01, max_iter=10, min_planes=20)     # warm (allocations)
    t0 = time.perf_counter()
    out, lg = c.damping_iter(P, form=0, u0=0.01, max_iter=10, min_planes=20)
    dt = time.perf_counter() - t0
    assert len(lg) == len(g["ref_log"])
    assert np.allclose(lg[:, :2], g["ref_log"][:, :2], rtol=1e-9, atol=2e-6)
    rot, tr = pose_errors(out, g["ref_poses"])
    print("real-world window: %d LM iterations in %.2f ms on the GPU (reference CPU: %.2f s); max pose diff %.2e rad %.2e m"
          % (len(lg), dt * 1e3, float(g["ref_seconds_lm"]), rot.max(), tr.max()))
    assert rot.max() <= ROT_TOL_RAD and tr.max() <= TRANS_TOL_M
    c.close()


def _two_rank_worker(rank, world, port, seed, W, F, pts, drop, q):
    import torch.distributed as dist
    from balm_amd import dist as bdist
    _os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
    bdist.init_process_group("gloo")          # two ranks share the one GPU of the box: RCCL refuses that, gloo does not
    sc, _ = make_scene(seed, W, F, pts, drop)
    nobs = (sc.clusters[..., 9] > 0).sum(1)
    lo, hi = bdist.partition_features(nobs, world)[rank]
    c = capi.Context(W, 0)
    c.set_features(sc.clusters[lo:hi], None, sc.coeffs[lo:hi])
    hook = bdist.install_allreduce(c)
    H, g, r = c.evaluate(0, sc.poses_init)
    out, lg = c.damping_iter(sc.poses_init, u0=0.01, max_iter=10)
    if rank == 0:
        q.put((H, g, r, out, lg, hook.zero_copy))
    dist.barrier()
    c.close()
    dist.destroy_process_group()


def test_two_ranks_sharded_on_one_gpu():
    """the N>1 path end to end on the GPU: two processes, each with its own feature shard and context,
    summing the device payload through the balm_set_allreduce hook; results must equal the
    single-process run (feature sums are order-insensitive to ~1e-13)."""
    import socket
    import torch.multiprocessing as mp
    seed, W, F, pts, drop = 81, 24, 120, 6, 0.25
    sc, _ = make_scene(seed, W, F, pts, drop)
    c = ctx_for(sc)
    H1, g1, r1 = c.evaluate(0, sc.poses_init)
    out1, lg1 = c.damping_iter(sc.poses_init, u0=0.01, max_iter=10)
    c.close()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, seed, W, F, pts, drop, q)) for r in range(2)]
    for p in procs:
        p.start()
    H2, g2, r2, out2, lg2, zc = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert rel_err(H2, H1) < 1e-12 and rel_err(g2, g1) < 1e-12 and abs(r2 - r1) / r1 < 1e-13
    assert len(lg2) == len(lg1) and np.allclose(lg2[:, :3], lg1[:, :3], rtol=1e-9)
    rot, tr = pose_errors(out2, out1)
    assert rot.max() < 1e-9 and tr.max() < 1e-9


def test_realworld_driver_end_to_end(tmp_path):
    """N2 product pipeline on the GPU box: scans + pose CSV in the shipped formats -> readers ->
    association (host C++) -> C ABI -> HIP LM loop; the oracle optimises the same features on the CPU."""
    from balm_amd import realworld as rw
    from oracle import assoc_host as ah
    from test_association import synthetic_window, write_window
    poses, frames = synthetic_window(1, 20, 150, 40)
    write_window(str(tmp_path), poses, frames)
    P, fr = rw.load_window(str(tmp_path))
    cl, co, layer = ah.associate(fr, P, 1.0)
    assert cl.shape[0] >= 3 * 20                                     # benchmark_realworld.cpp:209
    c = capi.Context(20)
    c.set_features(cl, None, co)
    out, lg = c.damping_iter(P, form=0, u0=0.01, max_iter=10, min_planes=20)
    oo, lo = orc.damping_iter(0, cl, None, co, P, 0.01, 10)
    assert len(lg) == len(lo)
    rot, tr = pose_errors(out, oo)
    assert rot.max() <= ROT_TOL_RAD and tr.max() <= TRANS_TOL_M
    assert lg[-1, 1] < lg[0, 0]
    c.close()
    assert rw.main([str(tmp_path), "--voxel", "1.0"]) == 0


def test_edge_cases_single_feature_two_observers_and_blind_pose():
    """domain edge cases: one feature only; a feature seen by exactly two poses (push_voxel's minimum,
    bavoxel.hpp:32-37); a pose that observes nothing (its 6x6 block of H and D is exactly zero: the
    damped matrix is singular, Eigen's LDLT leaves a zero pivot and its solve returns 0 there)."""
    sc, _ = make_scene(95, 9, 6, 25)
    sc.clusters[:, 4] = 0.0                     # pose 4 is blind
    sc.clusters[0, 2:] = 0.0                    # feature 0: poses 0 and 1 only
    sc.coeffs[:] = sc.clusters[..., 9].sum(1)
    for F in (1, 6):
        cl, co = sc.clusters[:F], sc.coeffs[:F]
        c = capi.Context(sc.W)
        c.set_features(cl, None, co)
        H, g, r = c.evaluate(0, sc.poses_init)
        Ho, go, ro = orc.evaluate(0, cl, None, co, sc.poses_init)
        assert abs(r - ro) / ro < 1e-12 and rel_err(g, go) < HTOL and rel_err(H, Ho) < HTOL
        assert not H[24:30].any() and not g[24:30].any()            # blind pose: exact zeros
        dx, q1 = c.solve_damped(Ho, go, 0.1)
        dxo, q1o = orc.solve_damped(Ho, go, 0.1)
        assert np.all(dx[24:30] == 0.0) and np.all(dxo[24:30] == 0.0)
        live = np.abs(dxo) > 0
        if F == 6:                                                   # F = 1 leaves the system rank deficient everywhere
            assert rel_err(dx[live], dxo[live]) < 1e-6
        c.close()


def test_virtual_driver_prints_the_reference_lines(capfd):
    """python -m balm_amd.virtual: the benchmark_virtual flow (generate, perturb, cluster build on the device, LM with
    that driver's constants) with the launch file's default sizes; RSME as the reference reports it"""
    from balm_amd import virtual
    assert virtual.main(["--seed", "5"]) == 0
    out = capfd.readouterr().out
    assert "winSize: 20" in out and "RSME:" in out and "iter0:" in out
    deg, m = [float(x.rstrip("degm,")) for x in out.split("RSME:")[1].split()[:2]]
    assert deg < 0.5 and m < 0.02           # 5 cm point noise: the reference lands in the same range


@pytest.mark.parametrize("gap", [1e-2, 1e-6, 1e-10, 1e-14, 0.0])
def test_device_eigen_matches_lapack_near_degenerate(gap):
    """The 3x3 symmetric eigen-decomposition on the device (Jacobi, k_feature_eigen; the stand-in for Eigen's
    SelfAdjointEigenSolver, bavoxel.hpp:345-351) against LAPACK (oracle/numpy_oracle.py uses numpy.linalg.eigh) on
    planes whose two in-plane eigenvalues are nearly or exactly equal: lambda_1 = lambda_2 (1 + gap).  The residual is
    coe * lambda_0 itself; H and g depend on u_1, u_2 only through the symmetric sum over k = 1, 2, which is well
    conditioned at the degeneracy although the individual eigenvectors are not."""
    from oracle import numpy_oracle as npo
    rng = np.random.default_rng(12)
    W, F, npt = 5, 24, 64
    sc = scene.generate(9, W, F, 8)
    clusters = np.zeros((F, W, 10))
    for a in range(F):
        # an isotropic disc (lambda_1 == lambda_2 by construction) stretched by (1 + gap) along one in-plane axis, a thin
        # normal direction, random orientation and offset; every pose sees the same world-frame points
        Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        ang = rng.uniform(0, 2 * np.pi, npt)
        rad = np.sqrt(rng.uniform(0, 1, npt))
        loc = np.stack([rad * np.cos(ang), rad * np.sin(ang), 1e-3 * rng.standard_normal(npt)], 1)
        loc -= loc.mean(0)
        C = loc[:, :2].T @ loc[:, :2] / npt
        w, V = np.linalg.eigh(C)
        loc[:, :2] = (loc[:, :2] @ V) / np.sqrt(w) * np.sqrt([1.0, 1.0 + gap])      # exact in-plane covariance diag(1, 1+gap)
        pw = loc @ Q.T + rng.uniform(-2, 2, 3)
        for i in range(W):
            R, p = npo.pose_R(sc.poses_gt)[i], npo.pose_p(sc.poses_gt)[i]
            pb = (pw[i::W] - p) @ R                                                   # body frame of pose i: R^T (q - p)
            clusters[a, i] = orc.cluster_push(pb.astype(np.float64))
    coeffs = clusters[..., 9].sum(1)
    c = capi.Context(W)
    c.set_features(clusters, None, coeffs)
    for poses in (sc.poses_gt, sc.poses_init):
        H, g, r = c.evaluate(0, poses)
        Hn, gn, rn = npo.left_evaluate(clusters, None, coeffs, poses)
        # lambda_0 ~ 1e-6 of lambda_1: an absolute error of one ulp of the matrix norm is ~1e-10 of the residual
        assert abs(r - rn) <= 1e-9 * abs(rn)
        assert rel_err(g, gn) < 1e-10 and rel_err(H, Hn) < 1e-12
    c.close()


@pytest.mark.parametrize("W,F,form", [(20, 60, 0), (100, 300, 1), (200, 2000, 0), (230, 40, 0), (300, 64, 1), (475, 24, 0), (700, 10, 0)])
def test_factor_kernel_store_paths_agree(W, F, form, monkeypatch):
    """k_feature_factors' two ways of writing Gt -- lane by lane (48 bytes per pose and column) and, the default where the LDS has room,
    coalesced through a staging block per wavefront -- must leave bit for bit the same H, g and residual (bavoxel.hpp:365-418 is what both
    evaluate); window sizes with one, two and four wavefronts of poses, a last wavefront that is partly empty, a window whose accumulators
    leave no room for the staging (475) and a pose-chunked one (700)"""
    sc, _ = make_scene(300 + W, W, F, 4, drop=0.25, mode=1)
    out = []
    for stage in ("0", "1"):
        monkeypatch.setenv("BALM_FACTORS_STAGE", stage)
        c = ctx_for(sc)
        out.append(c.evaluate(form, sc.poses_init))
        c.close()
    (H0, g0, r0), (H1, g1, r1) = out
    assert np.array_equal(H0, H1) and np.array_equal(g0, g1) and r0 == r1


@pytest.mark.parametrize("W,F", [(20, 60), (64, 700), (100, 300), (200, 2000), (256, 40)])
def test_factor_kernel_accumulator_homes_agree(W, F, monkeypatch):
    """k_feature_factors keeps a lane's pose and its 27 accumulators (gradient + block-diagonal sums, bavoxel.hpp:404-418) in registers for the left
    form on windows of up to 256 poses, in LDS otherwise (BALM_FACTORS_REGS=0): the same additions in the same order -> bit for bit the same
    H, g and residual; with the coalesced stores forced on and off"""
    sc, _ = make_scene(400 + W, W, F, 4, drop=0.25, mode=1)
    out = []
    for regs, stage in (("0", "0"), ("1", "0"), ("0", "1"), ("1", "1")):
        monkeypatch.setenv("BALM_FACTORS_REGS", regs)
        monkeypatch.setenv("BALM_FACTORS_STAGE", stage)
        c = ctx_for(sc)
        out.append(c.evaluate(0, sc.poses_init))
        c.close()
    for H, g, r in out[1:]:
        assert np.array_equal(H, out[0][0]) and np.array_equal(g, out[0][1]) and r == out[0][2]


@pytest.mark.parametrize("W,F,form", [(20, 150, 0), (20, 3000, 0), (24, 1000, 1), (7, 40, 0), (30, 700, 0)])
def test_tile_reduction_with_four_lanes_per_element_is_the_one_lane_sum(W, F, form, monkeypatch):
    """k_reduce_all sums the split-K partial tiles of a small window with four lanes per element (one per chain of reduce_tiles, the tail on
    chain 0, (s0 + s1) + (s2 + s3) by shuffles) instead of one thread walking up to 341 partials: the same additions in the same order -> bit
    for bit the same H (bavoxel.hpp:404-418); sizes whose slice counts are and are not multiples of four"""
    sc, _ = make_scene(900 + W, W, F, 4, drop=0.2, mode=1)
    monkeypatch.setenv("BALM_SYRK", "dense")
    out = []
    for q in ("0", "1"):
        monkeypatch.setenv("BALM_REDUCE_QUADS", q)
        c = ctx_for(sc)
        out.append(c.evaluate(form, sc.poses_init))
        out.append(c.evaluate(form, sc.poses_init, 3, F - 5))
        c.close()
    for k in (0, 1):
        assert np.array_equal(out[k][0], out[2 + k][0]) and np.array_equal(out[k][1], out[2 + k][1]) and out[k][2] == out[2 + k][2]
