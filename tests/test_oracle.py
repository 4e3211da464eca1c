"""CPU tests of the oracle itself (no GPU): the restatement is pinned by finite differences, the
rank-3 + block-diagonal identity, the right<->left adjoint relation and an independent numpy twin,
because the reference ships no golden vectors."""
import numpy as np
import pytest

from balm_amd import scene
from oracle import numpy_oracle as npo
from oracle import orc
from util import make_scene, rel_err


@pytest.fixture(scope="module")
def small():
    sc, fix = make_scene(7, 6, 5, 30, drop=0.25, with_fix=True)
    return sc, fix


def test_exp_log_roundtrip():
    rng = np.random.default_rng(0)
    for _ in range(20):
        w = rng.normal(size=3) * 0.7
        R = orc.exp(w)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14)
        assert np.allclose(orc.log(R), w, atol=1e-12)
        assert np.allclose(R, npo.exp_so3(w), atol=1e-15)
    assert np.array_equal(orc.exp(np.array([1e-12, 0, 0])), np.eye(3))   # tools.hpp:60 threshold


def test_eig3_matches_lapack():
    rng = np.random.default_rng(1)
    for _ in range(50):
        A = rng.normal(size=(3, 3)); A = A @ A.T
        A[2] *= 1e-3; A[:, 2] *= 1e-3
        lam, U = orc.eig3(A)
        ref = np.linalg.eigvalsh(A)
        assert np.allclose(lam, ref, rtol=1e-12, atol=1e-15 * np.abs(ref).max())
        assert np.allclose(A @ U, U * lam, atol=1e-13 * np.abs(A).max())


def test_cluster_push_and_transform():
    rng = np.random.default_rng(2)
    pts = rng.normal(size=(40, 3))
    cl = orc.cluster_push(pts)
    assert cl[9] == 40
    assert np.allclose(cl[6:9], pts.sum(0))
    P = pts.T @ pts
    assert np.allclose([cl[0], cl[1], cl[2], cl[3], cl[4], cl[5]], [P[0, 0], P[0, 1], P[0, 2], P[1, 1], P[1, 2], P[2, 2]])


@pytest.mark.parametrize("form", [0, 1])
def test_oracle_matches_numpy_twin(small, form):
    sc, fix = small
    H, g, r = orc.evaluate(form, sc.clusters, fix, sc.coeffs, sc.poses_init)
    f = npo.left_evaluate if form == 0 else npo.right_evaluate
    Hn, gn, rn = f(sc.clusters, fix, sc.coeffs, sc.poses_init)
    assert rel_err(H, Hn) < 1e-13
    assert rel_err(g, gn) < 1e-13
    assert abs(r - rn) / rn < 1e-13
    assert abs(orc.only_residual(sc.clusters, fix, sc.coeffs, sc.poses_init) - r) / r < 1e-13


@pytest.mark.parametrize("form", [0, 1])
def test_gradient_and_hessian_vs_finite_differences(small, form):
    sc, fix = small
    H, g, _ = orc.evaluate(form, sc.clusters, fix, sc.coeffs, sc.poses_init)
    gfd = npo.fd_gradient(form, sc.clusters, fix, sc.coeffs, sc.poses_init)
    assert rel_err(g, gfd) < 1e-7
    Hfd = npo.fd_hessian(form, sc.clusters, fix, sc.coeffs, sc.poses_init)
    assert rel_err(H, Hfd) < 2e-6


def test_rank3_blockdiag_identity(small):
    sc, fix = small
    H, _, _ = orc.evaluate(0, sc.clusters, fix, sc.coeffs, sc.poses_init)
    _, _, _, Gt, Bd = npo.left_evaluate(sc.clusters, fix, sc.coeffs, sc.poses_init, return_factors=True)
    Hb = np.zeros_like(H)
    for i in range(sc.W):
        Hb[6 * i:6 * i + 6, 6 * i:6 * i + 6] = Bd[i]
    assert rel_err(H, Hb - Gt @ Gt.T) < 1e-13


def test_right_left_adjoint_relation(small):
    sc, fix = small
    P = sc.poses_init
    _, gl, _ = orc.evaluate(0, sc.clusters, fix, sc.coeffs, P)
    _, gr, _ = orc.evaluate(1, sc.clusters, fix, sc.coeffs, P)
    R, p = npo.pose_R(P), npo.pose_p(P)
    n = 6 * sc.W
    LL = np.zeros((n, n))
    for i in range(sc.W):
        LL[6 * i:6 * i + 3, 6 * i:6 * i + 3] = R[i]
        LL[6 * i + 3:6 * i + 6, 6 * i:6 * i + 3] = npo.hat(p[i]) @ R[i]
        LL[6 * i + 3:6 * i + 6, 6 * i + 3:6 * i + 6] = np.eye(3)
    assert rel_err(gr, LL.T @ gl) < 1e-13


def test_thread_split_equals_single(small):
    sc, fix = small
    H1, g1, r1 = orc.evaluate(0, sc.clusters, fix, sc.coeffs, sc.poses_init)
    H4, g4, r4 = orc.evaluate_threads(0, sc.clusters, fix, sc.coeffs, sc.poses_init, 4)
    assert rel_err(H4, H1) < 1e-13 and rel_err(g4, g1) < 1e-13 and abs(r4 - r1) / r1 < 1e-13
    # sub-ranges add up (bavoxel.hpp:1049-1056)
    Ha, ga, ra = orc.evaluate(0, sc.clusters, fix, sc.coeffs, sc.poses_init, 0, 2)
    Hb, gb, rb = orc.evaluate(0, sc.clusters, fix, sc.coeffs, sc.poses_init, 2, sc.F)
    assert rel_err(Ha + Hb, H1) < 1e-13 and abs(ra + rb - r1) / r1 < 1e-13


def test_ldlt_indefinite_and_pd(small):
    sc, fix = small
    H, g, _ = orc.evaluate(0, sc.clusters, fix, sc.coeffs, sc.poses_init)
    for u in (0.01, 0.1, 10.0):
        A = H + u * np.diag(np.diag(H))
        x, neg = orc.ldlt_solve(A, -g)
        assert np.linalg.norm(A @ x + g) / np.linalg.norm(g) < 1e-10
        assert neg == int((np.linalg.eigvalsh(A) < 0).sum())     # Sylvester's law of inertia
        dx, q1 = orc.solve_damped(H, g, u)
        assert np.allclose(dx, x)
        assert np.isclose(q1, 0.5 * dx @ (u * np.diag(H) * dx - g))


def test_generator_is_deterministic_and_float_rounded():
    a = scene.generate(3, 5, 4, 6, keep_points=True)
    b = scene.generate(3, 5, 4, 6, keep_points=True)
    assert np.array_equal(a.clusters, b.clusters) and np.array_equal(a.poses_init, b.poses_init)
    # clusters are exactly the push of the float32 points (benchmark_virtual.cpp:397-402)
    for f in range(4):
        for i in range(5):
            cl = orc.cluster_push(a.points[f, i].astype(np.float64))
            assert np.allclose(cl, a.clusters[f, i], rtol=1e-15, atol=0)
    assert np.all(a.coeffs == 5 * 6)
    assert np.array_equal(a.poses_gt[0], np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0.0]))


@pytest.mark.parametrize("form,u0,max_iter", [(0, 0.1, 20), (0, 0.01, 10), (1, 0.1, 20)])
def test_lm_converges_to_ground_truth(form, u0, max_iter):
    sc = scene.generate(1, 20, 20, 40)
    out, lg = orc.damping_iter(form, sc.clusters, None, sc.coeffs, sc.poses_init, u0, max_iter)
    assert 3 <= len(lg) <= max_iter
    assert lg[-1, 1] < 0.05 * lg[0, 0]
    rot, tr = orc.rsme(orc.reanchor(sc.poses_gt), out)
    rot0, tr0 = orc.rsme(sc.poses_gt, sc.poses_init)
    assert rot * 57.3 < 0.2 and tr < 0.005          # ~0.06 deg, ~1.4 mm at config 1
    assert rot < 0.1 * rot0 and tr < 0.1 * tr0
