"""Block-sparse hessian_syrk plan (real co-visibility): must reproduce the dense plan and the oracle on windows whose
features see only a stretch of the trajectory, on both forms, and leave dense scenes on the dense plan."""
import os

import numpy as np
import pytest

from balm_amd import capi, scene
from oracle import orc
from util import rel_err

pytestmark = pytest.mark.gpu


def banded_scene(seed, W, F, pts, half_lo, half_hi, revisit=0.1):
    """every feature is seen from a contiguous stretch of poses (+ an occasional revisit elsewhere), as a lidar window's
    voxels are; weights = sum of N as VOX_HESS::push_voxel computes them (bavoxel.hpp:42-44)"""
    sc = scene.generate(seed, W, F, pts, mode=1)
    rng = np.random.default_rng(seed)
    keep = np.zeros((F, W), dtype=bool)
    centre = rng.integers(0, W, F)
    half = rng.integers(half_lo, half_hi + 1, F)
    idx = np.arange(W)[None, :]
    keep |= np.abs(idx - centre[:, None]) <= half[:, None]
    again = rng.uniform(size=F) < revisit
    c2 = rng.integers(0, W, F)
    keep |= again[:, None] & (np.abs(idx - c2[:, None]) <= 3)
    sc.clusters[~keep] = 0.0
    sc.coeffs[:] = sc.clusters[..., 9].sum(1)
    return sc


def context(sc, mode):
    if mode:
        os.environ["BALM_SYRK"] = mode
    else:
        os.environ.pop("BALM_SYRK", None)
    try:
        c = capi.Context(sc.W)
        c.set_features(sc.clusters, None, sc.coeffs)
    finally:
        os.environ.pop("BALM_SYRK", None)
    return c


@pytest.mark.parametrize("W,F", [(60, 700), (177, 2281), (300, 9000)])
def test_sparse_plan_matches_dense_plan_and_oracle(W, F):
    sc = banded_scene(W, W, F, 5, 4, 25)
    d, s, auto = context(sc, "dense"), context(sc, "sparse"), context(sc, None)
    wd, ws, wa = d.work_model(), s.work_model(), auto.work_model()
    assert ws["syrk_flops_issued"] < 0.7 * wd["syrk_flops_issued"]          # the plan really skips tiles
    assert wa["syrk_flops_issued"] == ws["syrk_flops_issued"]                # ... and is what the library picks by itself
    assert wd["syrk_flops_algorithmic"] == ws["syrk_flops_algorithmic"] == 216.0 * ws["B"]
    for form in (0, 1):
        Hd, gd, rd = d.evaluate(form, sc.poses_init)
        Hs, gs, rs = s.evaluate(form, sc.poses_init)
        assert rel_err(Hs, Hd) < 1e-12 and rel_err(gs, gd) < 1e-12 and abs(rs - rd) / rd < 1e-13
        assert form == 1 or np.array_equal(Hs, Hs.T)       # the left form's block diagonal is assembled from symmetric sums
    if F <= 2500:
        Ho, go, ro = orc.evaluate_threads(0, sc.clusters, None, sc.coeffs, sc.poses_init, 8)
        Hs, gs, rs = s.evaluate(0, sc.poses_init)
        assert rel_err(Hs, Ho) < 1e-10 and rel_err(gs, go) < 1e-10 and abs(rs - ro) / ro < 1e-12
    # sub-ranges (the reference's thread split) still add up to the whole
    cut = F // 3
    H1, g1, r1 = s.evaluate(0, sc.poses_init, 0, cut)
    H2, g2, r2 = s.evaluate(0, sc.poses_init, cut, F)
    Hs, gs, rs = s.evaluate(0, sc.poses_init)
    assert rel_err(H1 + H2, Hs) < 1e-12 and rel_err(g1 + g2, gs) < 1e-12
    # the LM run is the same run
    pd, ld = d.damping_iter(sc.poses_init, u0=0.01, max_iter=10)
    ps, ls = s.damping_iter(sc.poses_init, u0=0.01, max_iter=10)
    assert len(ld) == len(ls) and np.allclose(ld[:, :2], ls[:, :2], rtol=1e-9, atol=0) and np.abs(pd - ps).max() < 1e-9
    for c in (d, s, auto):
        c.close()


def test_dense_scene_keeps_the_dense_plan():
    sc = scene.generate(3, 64, 800, 6, mode=1)
    a, d = context(sc, None), context(sc, "dense")
    assert a.work_model() == d.work_model()
    a.close(); d.close()


def test_sparse_plan_in_shards():
    sc = banded_scene(11, 90, 1500, 5, 4, 20)
    a = context(sc, None)
    b = capi.Context(sc.W, 0, capi.FLAG_LOOPBACK_SHARDS, n_devices=3)
    b.set_features(sc.clusters, None, sc.coeffs)
    Ha, ga, ra = a.evaluate(0, sc.poses_init)
    Hb, gb, rb = b.evaluate(0, sc.poses_init)
    assert rel_err(Hb, Ha) < 1e-12 and rel_err(gb, ga) < 1e-12 and abs(ra - rb) / ra < 1e-13
    a.close(); b.close()
