"""Pins the oracle (CPU, no GPU) against the golden fixtures in tests/golden/ -- outputs of the
reference's own source compiled here (tests/golden/make_golden.py) -- and, where oracle/_ref is
available, directly against that build on fresh scenes.  The same fixtures gate the HIP path in
test_gpu_parity.py."""
import glob
import os

import numpy as np
import pytest

from oracle import orc, ref
from util import make_scene, pose_errors, rel_err

GOLD = sorted(g for g in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
              if not os.path.basename(g).startswith(("cov_", "assoc_", "lm_big_", "window_")))      # covariance fixtures: tests/test_cov_oracle.py; LM runs at the BASELINE sizes: tests/test_north_star.py


def load(path):
    return dict(np.load(path))


def test_fixtures_exist():
    assert len(GOLD) >= 3


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_matches_reference_fixtures(path):
    g = load(path)
    cl, co, P = g["clusters"], g["coeffs"], g["poses"]
    for form in (0, 1):
        H, J, r = orc.evaluate(form, cl, None, co, P)
        assert abs(r - g["r%d" % form]) / g["r%d" % form] < 1e-13
        assert rel_err(J, g["g%d" % form]) < 1e-12
        if "H%d" % form in g:
            assert rel_err(H, g["H%d" % form]) < 1e-12
    if "H2" in g:   # the reference's un-accelerated left form is the same matrix
        assert rel_err(orc.evaluate(0, cl, None, co, P)[0], g["H2"]) < 1e-12
    assert abs(orc.only_residual(cl, None, co, P) - g["r_only"]) / g["r_only"] < 1e-13
    for u in (0.01, 0.1):
        dx, q1 = orc.solve_damped(g["H0"], g["g0"], u)
        assert rel_err(dx, g["dx_u%g" % u]) < 1e-9
        assert abs(q1 - g["q1_u%g" % u]) / abs(g["q1_u%g" % u]) < 1e-10
    if "lm_poses" in g:
        out, lg = orc.damping_iter(0, cl, None, co, P, 0.01, 10)
        assert len(lg) == len(g["lm_log"])
        big = np.abs(lg[:, 4]) > 1e-5            # the reference prints q with 6 decimals: sign lost below that
        assert np.array_equal(lg[big, 6], g["lm_log"][big, 6])
        assert np.allclose(lg[:, :2], g["lm_log"][:, :2], atol=2e-6)      # printf("%lf") rounding
        rot, tr = pose_errors(out, g["lm_poses"])
        assert rot.max() < 1e-9 and tr.max() < 1e-9


needs_ref = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")


@needs_ref
@pytest.mark.parametrize("seed,W,F,pts,drop", [(21, 12, 30, 10, 0.0), (22, 25, 40, 6, 0.4), (23, 5, 3, 50, 0.0)])
def test_oracle_matches_compiled_reference(seed, W, F, pts, drop):
    sc, _ = make_scene(seed, W, F, pts, drop)
    for form in (0, 1):
        Hr, gr, rr = ref.evaluate(form, sc.clusters, None, sc.coeffs, sc.poses_init)
        Ho, go, ro = orc.evaluate(form, sc.clusters, None, sc.coeffs, sc.poses_init)
        assert rel_err(Ho, Hr) < 1e-13 and rel_err(go, gr) < 1e-13 and abs(ro - rr) / rr < 1e-13
    # sub-range + 4-thread split of the reference (bavoxel.hpp:1025-1059)
    Hr, gr, rr = ref.divide_thread(0, sc.clusters, None, sc.coeffs, sc.poses_init)
    Ho, go, ro = orc.evaluate_threads(0, sc.clusters, None, sc.coeffs, sc.poses_init, 4)
    assert rel_err(Ho, Hr) < 1e-13 and abs(ro - rr) / rr < 1e-13
    assert abs(ref.only_residual(sc.clusters, None, sc.coeffs, sc.poses_init)
               - orc.only_residual(sc.clusters, None, sc.coeffs, sc.poses_init)) / ro < 1e-13


@needs_ref
def test_exp_log_match_reference_tools_hpp():
    rng = np.random.default_rng(5)
    for _ in range(20):
        w = rng.normal(size=3) * rng.choice([1e-12, 1e-3, 1.0, 3.0])
        assert np.array_equal(orc.exp(w), ref.exp(w))
        R = ref.exp(w)
        assert np.allclose(orc.log(R), ref.log(R), atol=1e-15)


@needs_ref
def test_push_voxel_weight_and_filter():
    sc, _ = make_scene(24, 8, 6, 5, drop=0.5)
    for a in range(sc.F):
        kept, coe = ref.push_voxel(sc.clusters[a])
        nobs = int((sc.clusters[a, :, 9] > 0).sum())
        assert kept == (nobs >= 2)                                  # bavoxel.hpp:32-37
        if kept:
            assert coe == sc.clusters[a, :, 9].sum() == sc.coeffs[a]   # :42-44 (scene.sparsify mirrors it)


def test_oracle_matches_reference_on_realworld_window():
    """the shipped real-world window (if the fixture was built): oracle vs the reference's optimizer"""
    from conftest import ROOT
    path = os.path.join(ROOT, "oracle", "_ref", "realworld_features.npz")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/realworld_features.npz not built")
    g = dict(np.load(path))
    out, lg = orc.damping_iter(0, g["clusters"], None, g["coeffs"], g["poses"], 0.01, 10, threads=8)
    assert len(lg) == len(g["ref_log"])
    assert np.allclose(lg[:, :2], g["ref_log"][:, :2], rtol=1e-9, atol=2e-6)
    rot, tr = pose_errors(out, g["ref_poses"])
    assert rot.max() < 1e-9 and tr.max() < 1e-9
