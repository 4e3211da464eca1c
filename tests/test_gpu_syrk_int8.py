"""BALM_SYRK=int8 (opt-in, round 6): the Hessian's Gt Gt^T (bavoxel.hpp:404-418, K3 of DESIGN.md) on the INT8 matrix cores by error-free
digit slicing (balm_amd/csrc/kernels_syrk_i8.hip) against the default FP64 path and against the reference's own values.

The contract of the switch: g and the residual do not pass through the product and stay bit-identical; an entry of H carries an
error of about 2^-32 of (largest |entry| of row i of Gt) x (of row j) per column, unsigned -- 1.4e-12 of the largest entry at BASELINE
configs[2], up to 1.3e-10 on windows of a few hundred columns (no averaging).  By itself the switch engages from 12 288 columns and 96 poses on.  The default path, its tolerances and the bench line stay FP64."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balm_amd import scene  # noqa: E402

pytestmark = pytest.mark.gpu


def evaluate(sc, mode, form=0, fix=None, sub=None):
    from balm_amd import capi
    if mode:
        os.environ["BALM_SYRK"] = mode
    else:
        os.environ.pop("BALM_SYRK", None)
    os.environ["BALM_SYRK_INT8_MIN_COLS"] = "0"      # (the switch engages from 12 288 columns on by itself: here at every size)
    try:
        c = capi.Context(sc.W)
        c.set_features(sc.clusters, fix, sc.coeffs)
        out = c.evaluate(form, sc.poses_init) if sub is None else c.evaluate(form, sc.poses_init, sub[0], sub[1])
        c.close()
    finally:
        os.environ.pop("BALM_SYRK", None)
        os.environ.pop("BALM_SYRK_INT8_MIN_COLS", None)
    return out


@pytest.mark.parametrize("seed,W,F,form", [(5, 20, 60, 0), (6, 33, 500, 0), (7, 100, 3000, 0), (8, 213, 1000, 0), (9, 7, 9, 0),
                                           (10, 64, 700, 1), (11, 300, 400, 0), (12, 475, 64, 0), (13, 8, 90000, 0)])
def test_int8_product_against_the_fp64_product(seed, W, F, form):
    """ragged windows (n = 42 < one 128-row tile; 213 poses: a padded last tile), both forms (the right form and windows above 256 poses
    take the row maxima from a pass over Gt, the others from the factor kernel), half-empty tables; 270 000 columns: two k-slices per XCD
    (a slice's int32 sums hold 32 704 columns of four digit pairs)"""
    sc = scene.generate(seed, W, F, 6, mode=1)
    scene.sparsify(sc, seed + 100, 0.3)
    Hd, gd, rd = evaluate(sc, "dense", form)
    Hi, gi, ri = evaluate(sc, "int8", form)
    scale = np.abs(np.diag(Hd)).max()
    assert np.array_equal(gd, gi) and rd == ri
    assert np.abs(Hi - Hd).max() <= 5e-10 * scale, np.abs(Hi - Hd).max() / scale
    assert form == 1 or np.array_equal(Hi, Hi.T)


def test_int8_engages_from_12288_columns_and_96_poses_on_by_itself(monkeypatch):
    """below its thresholds BALM_SYRK=int8 leaves the FP64 product in place -- bit for bit the default's Hessian: fewer than 12 288 columns (nothing
    to gain, nothing averaging the truncation), fewer than 96 poses (a handful of tiles for 256 CUs: the INT8 kernel is the slower one); above, not"""
    from balm_amd import capi
    out = {}
    for W, F in ((96, 4000), (96, 4200), (64, 6000)):
        sc = scene.generate(31, W, F, 6, mode=1)
        for mode in ("dense", "int8"):
            monkeypatch.setenv("BALM_SYRK", mode)
            c = capi.Context(sc.W)
            c.set_features(sc.clusters, None, sc.coeffs)
            out[W, F, mode] = c.evaluate(0, sc.poses_init)[0]
            c.close()
    assert np.array_equal(out[96, 4000, "dense"], out[96, 4000, "int8"])
    assert np.array_equal(out[64, 6000, "dense"], out[64, 6000, "int8"])
    d = np.abs(out[96, 4200, "dense"] - out[96, 4200, "int8"]).max()
    assert 0 < d <= 1e-10 * np.abs(np.diag(out[96, 4200, "dense"])).max()


@pytest.mark.parametrize("W,form", [(40, 0), (40, 1)])
def test_int8_product_keeps_a_non_finite_factor_non_finite(W, form):
    """a feature that is ONE point (eight copies of a pose's origin, seen by that pose only) has the covariance 0 exactly: lambda_1 - lambda_0 = 0,
    c_1 = sqrt(2 w / 0) / N = inf (bavoxel.hpp:371-378 divides the same way) and its columns of Gt are non-finite for that pose while g and the
    residual stay finite.  The FP64 product leaves non-finite rows and columns in H for that pose; so must the sliced one (a NaN dropped by the
    row maximum would slice the row to zeros) -- through the factor kernel's maxima (left form) and through the pass over Gt (right form)"""
    sc = scene.generate(41, W, 300, 6, mode=1)
    seen = np.array([17])
    sc.clusters = sc.clusters.copy()
    sc.clusters[123] = 0.0
    sc.clusters[123, 17, 9] = 8.0
    sc.coeffs = sc.clusters[..., 9].sum(1)
    Hd = evaluate(sc, "dense", form)[0]
    Hi = evaluate(sc, "int8", form)[0]
    assert len(seen) > 0 and not np.isfinite(Hd).all()
    rows = (6 * seen[:, None] + np.arange(6)[None, :]).reshape(-1)
    bad_d, bad_i = ~np.isfinite(Hd), ~np.isfinite(Hi)
    assert bad_d[rows].any(axis=1).all() and bad_i[rows].any(axis=1).all()
    assert np.array_equal(bad_i.any(axis=1), bad_d.any(axis=1))        # the same rows are touched


def test_int8_product_on_a_feature_sub_range_and_after_a_wider_one():
    """evaluate(head, end): the product over a sub-range of the columns (fewer k-steps, the same scratch)"""
    sc = scene.generate(21, 50, 900, 6, mode=1)
    Hd, _, _ = evaluate(sc, "dense", 0, None, (100, 640))
    Hi, _, _ = evaluate(sc, "int8", 0, None, (100, 640))
    assert np.abs(Hi - Hd).max() <= 5e-10 * np.abs(np.diag(Hd)).max()


def test_int8_full_size_hessian_against_the_reference_directly():
    """BASELINE configs[2] (W = 200, F = 50 000) in one evaluation against the reference's own divide_thread_left values
    (tests/golden/make_golden_eval.py), at a TENTH of the FP64 test's own tolerance: 1e-11 of the largest entry (measured: 1.4e-12, H V 6.1e-12)"""
    from test_north_star import GOLD, load
    sys.path.insert(0, GOLD)
    from make_golden_eval import probe_vectors
    g, sc = load("lm_big_w200_f50000")
    H, grad, r = evaluate(sc, "int8")
    scale = np.abs(g["eval_diag"]).max()
    V = probe_vectors(H.shape[0])
    assert abs(r - float(g["eval_r"])) <= 1e-12 * abs(float(g["eval_r"]))
    assert np.abs(grad - g["eval_g"]).max() <= 1e-10 * np.abs(g["eval_g"]).max()
    assert np.abs(np.diag(H) - g["eval_diag"]).max() <= 1e-11 * scale
    assert np.abs(H[::97] - g["eval_rows"]).max() <= 1e-11 * scale
    assert np.abs(H @ V - g["eval_HV"]).max() <= 1e-11 * np.abs(g["eval_HV"]).max()


@pytest.mark.parametrize("case,which", [("lm_big_w64_f5000", "bavoxel"), ("lm_big_w64_f5000", "virtual"), ("lm_big_w200_f50000", "bavoxel")])
def test_int8_lm_run_reproduces_the_reference_run(case, which, monkeypatch):
    """the north-star acceptance test under the switch: same iteration count, accept sequence, residuals to the printed decimals, final poses
    within 1e-5 rad / 1e-4 m of the reference's run (the trial evaluations' factor kernel supplies the row maxima here)"""
    from balm_amd import capi
    from test_north_star import CONSTANTS, check_run, load
    monkeypatch.setenv("BALM_SYRK", "int8")
    monkeypatch.setenv("BALM_SYRK_INT8_MIN_COLS", "0")
    k = CONSTANTS[which]
    g, sc = load(case)
    c = capi.Context(sc.W)
    c.set_features(sc.clusters, None, sc.coeffs)
    out, lg = c.damping_iter(sc.poses_init, form=0, u0=k["u0"], max_iter=k["max_iter"], min_planes=k["min_planes"])
    c.close()
    rot, tr = check_run(g, which, out, lg)
    print("%s %s under BALM_SYRK=int8: %d iterations, max pose difference to the reference %.2e rad %.2e m" % (case, which, len(lg), rot, tr))


def test_int8_pose_covariance_against_the_fp64_stage():
    """balm_pose_covariance (N4, benchmark consistency: Rcov = H^-1 (X X^T + Y Y^T + S) H^-T) with the Hessian's and the stage's two SYRKs on the
    INT8 product against the FP64 stage, at the smallest window the switch engages at by itself"""
    from balm_amd import capi
    sc = scene.generate(51, 96, 4200, 6, mode=1)
    fix = 0.3 * sc.clusters[:, 0]
    fix[:, 9] = np.round(fix[:, 9])
    out = {}
    for mode in ("dense", "int8"):
        os.environ["BALM_SYRK"] = mode
        try:
            c = capi.Context(sc.W)
            c.set_features(sc.clusters, fix, np.ones(sc.F))
            out[mode] = c.pose_covariance(sc.poses_init, point_sigma=0.02, want_raw=True)
            c.close()
        finally:
            os.environ.pop("BALM_SYRK", None)
    for a, b in zip(out["dense"], out["int8"]):
        assert np.abs(a - b).max() <= 1e-7 * np.abs(a).max() and not np.array_equal(a, b)      # (measured 3e-8: H^-1 . H^-T multiplies the Hessian's 1e-11 by its conditioning)


def test_int8_through_the_create_flag_is_the_environment_switch():
    """BALM_FLAG_SYRK_INT8 at balm_create (the production form of the opt-in; include/balm_hip.h) selects the same product as BALM_SYRK=int8:
    bit for bit the same Hessian, and not the FP64 one"""
    from balm_amd import capi
    sc = scene.generate(61, 100, 4200, 6, mode=1)
    He = evaluate(sc, "int8")[0]
    Hd = evaluate(sc, None)[0]
    c = capi.Context(sc.W, 0, capi.FLAG_SYRK_INT8)
    c.set_features(sc.clusters, None, sc.coeffs)
    Hf = c.evaluate(0, sc.poses_init)[0]
    c.close()
    assert np.array_equal(Hf, He) and not np.array_equal(Hf, Hd)
    assert np.abs(Hf - Hd).max() <= 1e-10 * np.abs(np.diag(Hd)).max()


def test_int8_product_inside_replayed_lm_graphs(monkeypatch):
    """BALM_GRAPH=1 captures the LM iteration -- with the switch: the slicing, the INT8 product and its packing among the captured launches -- and
    replays it: the same poses and the same trace as plain launches, bit for bit"""
    from balm_amd import capi
    sc = scene.generate(3, 20, 600, 6, mode=1)
    monkeypatch.setenv("BALM_SYRK", "int8")
    monkeypatch.setenv("BALM_SYRK_INT8_MIN_COLS", "0")
    out = []
    for g in ("0", "1"):
        monkeypatch.setenv("BALM_GRAPH", g)
        c = capi.Context(sc.W)
        c.set_features(sc.clusters, None, sc.coeffs)
        out.append(c.damping_iter(sc.poses_init, form=0, u0=0.1, max_iter=12, force_hess=True, no_stop=True, reanchor=False))
        c.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and len(out[1][1]) == 12


@pytest.mark.parametrize("form", [0, 1])
def test_int8_with_the_one_pass_trial_evaluation(form, monkeypatch):
    """BALM_FUSE_TRIAL=1: the trial evaluation's k_moments_factors leaves the factors of the trial poses in the second Gt buffer -- under the
    switch also their row maxima, which change hands with the buffer when the step is accepted.  The LM run must make the decisions of the
    three-kernel path under the switch and land on its poses; then a plain evaluation must still be right (no stale maxima)"""
    from balm_amd import capi
    sc = scene.generate(71, 48, 1500, 6, mode=1)
    scene.sparsify(sc, 171, 0.2)
    monkeypatch.setenv("BALM_SYRK", "int8")
    monkeypatch.setenv("BALM_SYRK_INT8_MIN_COLS", "0")
    c = capi.Context(sc.W)
    c.set_features(sc.clusters, None, sc.coeffs)
    monkeypatch.delenv("BALM_FUSE_TRIAL", raising=False)
    pa, la = c.damping_iter(sc.poses_init, form=form, u0=0.1, max_iter=8, min_planes=0)
    monkeypatch.setenv("BALM_FUSE_TRIAL", "1")
    pb, lb = c.damping_iter(sc.poses_init, form=form, u0=0.1, max_iter=8, min_planes=0)
    monkeypatch.delenv("BALM_FUSE_TRIAL")
    assert len(la) == len(lb) and np.array_equal(la[:, 6], lb[:, 6]) and la[:, 6].sum() >= 2
    assert np.allclose(la[:, :2], lb[:, :2], rtol=1e-9, atol=0) and np.abs(pa - pb).max() < 1e-8
    Hi = c.evaluate(form, sc.poses_init)[0]
    c.close()
    monkeypatch.setenv("BALM_SYRK", "dense")
    d = capi.Context(sc.W)
    d.set_features(sc.clusters, None, sc.coeffs)
    Hd = d.evaluate(form, sc.poses_init)[0]
    pd, ld = d.damping_iter(sc.poses_init, form=form, u0=0.1, max_iter=8, min_planes=0)
    d.close()
    assert np.abs(Hi - Hd).max() <= 5e-10 * np.abs(np.diag(Hd)).max()
    assert len(ld) == len(la) and np.abs(pd - pa).max() < 1e-8
