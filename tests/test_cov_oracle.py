"""N4, CPU side: the numpy restatement of the covariance path (oracle/numpy_oracle.py) against
golden vectors produced by the reference's own compiled sources (tests/golden/make_golden_cov.py) and, where
oracle/_ref is built, against that code directly.  Tolerances: relative to the largest entry, FP64 round-off of
O(W F) sums -- 1e-11 for Rcov_raw, 1e-9 for Rcov (it goes through H^-1 twice, cond(H) ~ 1e3)."""
import os

import numpy as np
import pytest

from balm_amd import scene
from conftest import ROOT
from oracle import numpy_oracle as npo
from oracle import orc, ref_sim

needs_ref = pytest.mark.skipif(not ref_sim.available(), reason="oracle/_ref/libbalm_ref_sim.so not built")


def golden():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "cov_w6_f10.npz")))


def rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def test_cluster_noise_covariance_closed_form_matches_golden():
    g = golden()
    cc = npo.cluster_noise_cov_closed_form(g["clusters"], float(g["pn"]))
    assert rel(cc, g["ccov"]) < 1e-14          # PointCluster::push accumulates the same sums point by point


def test_point_covariance_literal_and_factored_match_golden():
    g = golden()
    Rl = npo.point_cov_left(g["clusters"], g["ccov"], g["fix"], g["poses"])
    assert rel(Rl, g["Rraw"]) < 1e-11
    Rf, X, Y, S = npo.point_cov_left_factored(g["clusters"], g["ccov"], g["fix"], g["poses"])
    assert rel(Rf, g["Rraw"]) < 1e-11 and X.shape == (36, 30)
    assert np.abs(g["Rraw"] - g["Rraw"].T).max() < 1e-14 * np.abs(g["Rraw"]).max()


def test_pose_covariance_matches_golden():
    g = golden()
    H, _, _ = orc.evaluate(0, g["clusters"], g["fix"], np.ones(10), g["poses"])
    assert rel(H, g["Hess"]) < 1e-12           # the simulation's evaluator is the path's left evaluator, fix included
    Rf = npo.point_cov_left_factored(g["clusters"], g["ccov"], g["fix"], g["poses"])[0]
    assert rel(npo.pose_cov(H, Rf), g["Rcov"]) < 1e-9
    assert np.linalg.eigvalsh(g["Rcov"]).min() > 0


@needs_ref
@pytest.mark.parametrize("seed,W,F", [(3, 5, 7), (8, 12, 25)])
def test_against_compiled_reference(seed, W, F):
    sc = scene.generate(seed, W, F, 25)
    cl = sc.clusters.copy()
    cl[1, 2] = 0
    cl[F - 1, : W // 2] = 0
    cc = npo.cluster_noise_cov_closed_form(cl, 0.03)
    fix = 0.5 * cl[:, 0]
    fix[:, 9] = np.round(fix[:, 9])
    Rr = ref_sim.point_cov(cl, cc, fix, sc.poses_init)
    assert rel(npo.point_cov_left(cl, cc, fix, sc.poses_init), Rr) < 1e-11
    assert rel(npo.point_cov_left_factored(cl, cc, fix, sc.poses_init)[0], Rr) < 1e-11
    # feature sub-ranges add up (the reference splits them over four threads, BAs_left.hpp:1006-1010)
    h = F // 2
    assert rel(ref_sim.point_cov(cl, cc, fix, sc.poses_init, 0, h) + ref_sim.point_cov(cl, cc, fix, sc.poses_init, h, F), Rr) < 1e-12
    Hr, Rc = ref_sim.pose_cov(cl, cc, fix, sc.poses_init)
    assert rel(npo.pose_cov(Hr, npo.point_cov_left_factored(cl, cc, fix, sc.poses_init)[0]), Rc) < 1e-9


def test_weights_scale_like_the_gradient():
    g = golden()
    co = np.linspace(0.5, 2.0, 10)
    Rw = npo.point_cov_left(g["clusters"], g["ccov"], g["fix"], g["poses"], coeffs=co)
    Rf = npo.point_cov_left_factored(g["clusters"], g["ccov"], g["fix"], g["poses"], coeffs=co)[0]
    assert rel(Rf, Rw) < 1e-11
