"""The damped solve's two factorisation paths -- the persistent cooperative kernel (k_ldl_fused) and the per-panel launch
pair (k_ldl_panel + k_ldl_trail, kept for BALM_SOLVE=launches and as the path of devices without cooperative launch) --
must agree to rounding (same elimination order, pivot-block code and D^+ rule; W = L D is re-formed in the fused path), and both must agree with LAPACK."""
import os

import numpy as np
import pytest

from balm_amd import capi
from util import make_scene, rel_err

pytestmark = pytest.mark.gpu


def both_paths(c, H, g, u):
    os.environ["BALM_SOLVE"] = "launches"
    dx0, q0 = c.solve_damped(H, g, u)
    os.environ["BALM_SOLVE"] = "fused"
    dx1, q1 = c.solve_damped(H, g, u)
    os.environ.pop("BALM_SOLVE")
    return (dx0, q0), (dx1, q1)


@pytest.mark.parametrize("W", [8, 16, 17, 24, 33, 64, 100, 200])
@pytest.mark.parametrize("kind", ["spd", "indefinite"])
def test_fused_factorisation_matches_launch_pair(W, kind):
    rng = np.random.default_rng(W * 7 + (kind == "spd"))
    n = 6 * W
    B = rng.standard_normal((n, n))
    H = B @ B.T / n + np.diag(rng.uniform(0.5, 50.0, n))
    if kind == "indefinite":
        s = np.where(rng.uniform(size=n) < 0.2, -1.0, 1.0)
        H = (H * s[:, None]) * s[None, :]
        H[np.diag_indices(n)] *= s            # negative pivots on the diagonal, as the exact Hessian has away from the optimum
    g = rng.standard_normal(n)
    c = capi.Context(W)
    (dx0, q0), (dx1, q1) = both_paths(c, H, g, 0.1)
    assert rel_err(dx1, dx0) < 1e-11 and abs(q0 - q1) <= 1e-11 * abs(q0)
    D = np.diag(np.diag(H))
    ref = np.linalg.solve(H + 0.1 * D, -g)
    assert rel_err(dx1, ref) < 1e-9
    c.close()


def test_fused_path_on_a_real_hessian_and_lm_run():
    sc, _ = make_scene(21, 48, 500, 8, drop=0.3)
    c = capi.Context(sc.W)
    c.set_features(sc.clusters, None, sc.coeffs)
    H, g, _ = c.evaluate(0, sc.poses_init)
    for u in (0.01, 1.0):
        (dx0, q0), (dx1, q1) = both_paths(c, H, g, u)
        assert rel_err(dx1, dx0) < 1e-11 and abs(q0 - q1) <= 1e-11 * abs(q0)
    os.environ["BALM_SOLVE"] = "launches"
    pa, la = c.damping_iter(sc.poses_init, u0=0.01, max_iter=10)
    os.environ["BALM_SOLVE"] = "fused"
    pb, lb = c.damping_iter(sc.poses_init, u0=0.01, max_iter=10)
    os.environ.pop("BALM_SOLVE")
    assert len(la) == len(lb) and np.allclose(la[:, :3], lb[:, :3], rtol=1e-9, atol=0) and np.abs(pa - pb).max() < 1e-10
    c.close()


def test_fused_path_many_solves_in_a_row():
    """flags are re-zeroed per solve; stale flags from the previous factorisation would let a workgroup run ahead"""
    rng = np.random.default_rng(5)
    W = 40
    n = 6 * W
    c = capi.Context(W)
    os.environ["BALM_SOLVE"] = "fused"
    for k in range(25):
        B = rng.standard_normal((n, n))
        H = B @ B.T / n + np.diag(rng.uniform(0.5, 5.0, n))
        g = rng.standard_normal(n)
        dx, _ = c.solve_damped(H, g, 0.05)
        ref = np.linalg.solve(H + 0.05 * np.diag(np.diag(H)), -g)
        assert rel_err(dx, ref) < 1e-9, k
    os.environ.pop("BALM_SOLVE")
    c.close()


@pytest.mark.parametrize("W", [8, 24, 100, 400])
def test_lookahead_launch_path(W):
    """the lookahead form of the launch path (default for windows above the persistent kernel's range: one launch per panel,
    the next panel's own columns updated in its workgroups' registers, the rest of the trailing update beside its steps)
    agrees with the plain launch pair to rounding, and with itself bit for bit solve after solve (a missing dependency
    would show up as run-to-run differences)"""
    rng = np.random.default_rng(W)
    n = 6 * W
    B = rng.standard_normal((n, 64))
    H = B @ B.T / 64 + np.diag(rng.uniform(0.5, 50.0, n))
    g = rng.standard_normal(n)
    c = capi.Context(W)
    os.environ["BALM_SOLVE"] = "launches"
    try:
        os.environ["BALM_LOOKAHEAD"] = "0"
        dx0, q0 = c.solve_damped(H, g, 0.1)
        os.environ["BALM_LOOKAHEAD"] = "1"
        dx1, q1 = c.solve_damped(H, g, 0.1)
        assert rel_err(dx1, dx0) < 1e-11 and abs(q0 - q1) <= 1e-11 * abs(q0)
        for _ in range(6):
            dx2, q2 = c.solve_damped(H, g, 0.1)
            assert np.array_equal(dx1, dx2) and q1 == q2
    finally:
        os.environ.pop("BALM_SOLVE"); os.environ.pop("BALM_LOOKAHEAD", None)
    c.close()


def _test_matrix(W, kind, seed):
    rng = np.random.default_rng(seed)
    n = 6 * W
    B = rng.standard_normal((n, 96))
    H = B @ B.T / 96 + np.diag(rng.uniform(0.5, 50.0, n))
    if kind == "indefinite":
        s = np.where(rng.uniform(size=n) < 0.2, -1.0, 1.0)
        H = (H * s[:, None]) * s[None, :]
        H[np.diag_indices(n)] *= s
    return H, rng.standard_normal(n)


@pytest.mark.parametrize("W", [350, 400, 500, 700, 1024])
@pytest.mark.parametrize("kind", ["spd", "indefinite"])
def test_large_window_paths_match_lapack(W, kind):
    """An INDEPENDENT comparator for the factorisation paths windows above 320 poses take (BASELINE configs[4] names
    W = 500): the launch path with lookahead (k_ldl_panel_trail, their default), the plain launch pair and the persistent
    kernel are each compared with LAPACK (numpy.linalg.solve of the damped system, bavoxel.hpp:1113-1114), not with one
    another; then with one another to rounding."""
    H, g = _test_matrix(W, kind, 1000 + W + (kind == "spd"))
    u = 0.1
    ref = np.linalg.solve(H + u * np.diag(np.diag(H)), -g)
    q1_ref = 0.5 * ref @ (u * np.diag(H) * ref - g)              # bavoxel.hpp:1127
    c = capi.Context(W)
    got = {}
    try:
        dx, q1 = c.solve_damped(H, g, u)                         # the default for this window size
        got["default"] = dx
        assert rel_err(dx, ref) < 1e-9 and abs(q1 - q1_ref) <= 1e-9 * abs(q1_ref), ("default", rel_err(dx, ref))
        for mode, la in (("launches", "1"), ("launches", "0"), ("fused", None)):
            os.environ["BALM_SOLVE"] = mode
            if la is not None:
                os.environ["BALM_LOOKAHEAD"] = la
            dx, q1 = c.solve_damped(H, g, u)
            os.environ.pop("BALM_LOOKAHEAD", None)
            got[mode + (la or "")] = dx
            assert rel_err(dx, ref) < 1e-9 and abs(q1 - q1_ref) <= 1e-9 * abs(q1_ref), (mode, la, rel_err(dx, ref))
    finally:
        os.environ.pop("BALM_SOLVE", None); os.environ.pop("BALM_LOOKAHEAD", None)
    for k, v in got.items():
        assert rel_err(v, got["default"]) < 1e-10, k
    c.close()


def test_large_window_solve_of_an_lm_hessian_matches_lapack():
    """the same at W = 500 on a Hessian of the path itself (sparse co-visibility, indefinite at the noisy start): residual
    of the damped system and LAPACK's solution"""
    sc, _ = make_scene(77, 500, 600, 6, drop=0.5)
    c = capi.Context(sc.W)
    c.set_features(sc.clusters, None, sc.coeffs)
    H, g, _ = c.evaluate(0, sc.poses_init)
    for u in (0.01, 0.1):
        A = H + u * np.diag(np.diag(H))
        ref = np.linalg.solve(A, -g)
        dx, _ = c.solve_damped(H, g, u)
        assert np.linalg.norm(A @ dx + g) / np.linalg.norm(g) < 1e-8
        assert rel_err(dx, ref) < 1e-7
    c.close()


@pytest.mark.parametrize("W", [40, 41, 64, 100, 200, 328, 336, 400, 500, 640, 700, 800])
@pytest.mark.parametrize("kind", ["spd", "indefinite"])
def test_chain_kernel_with_back_substitution_matches_lapack(W, kind):
    """k_ldl_chain on [A ; rhs] alone + k_ldl_backsolve (round 3: the default from 31 to 100 panels, i.e. 248 .. 800 poses;
    BALM_SOLVE=chainb forces it from 5 panels on) against LAPACK and against the launch path, solve after solve (the exchange
    buffer and the flags are re-armed per solve); a vanished pivot (indefinite case aside: an exactly singular block) is the
    pseudo-inverse in both"""
    H, g = _test_matrix(W, kind, 31 * W + (kind == "spd"))
    u = 0.1
    ref = np.linalg.solve(H + u * np.diag(np.diag(H)), -g)
    c = capi.Context(W)
    os.environ["BALM_SOLVE"] = "chainb"
    try:
        for _ in range(3):
            dx, q1 = c.solve_damped(H, g, u)
            assert np.all(np.isfinite(dx)), "k_ldl_chain / k_ldl_backsolve gave up on a flag (bounded waits) or was not launched"
            assert rel_err(dx, ref) < 1e-9
        os.environ["BALM_SOLVE"] = "launches"
        dx0, q0 = c.solve_damped(H, g, u)
        assert rel_err(dx, dx0) < 1e-10 and abs(q1 - q0) <= 1e-10 * abs(q0)
    finally:
        os.environ.pop("BALM_SOLVE", None)
    c.close()


@pytest.mark.parametrize("W", [40, 264])
def test_blind_poses_leave_zero_pivots_in_every_path(W):
    """poses that observe nothing have exactly zero 6x6 blocks in H and D: the damped matrix is singular, Eigen's LDLT leaves
    zero pivots and its solve returns 0 there (D^+).  Every factorisation path -- the product form with L^-T D^+ and the block
    back-substitution with Minv_b D_bb -- must return exact zeros for those poses and the pseudo-inverse solution elsewhere"""
    rng = np.random.default_rng(W)
    n = 6 * W
    blind = rng.choice(W, size=max(2, W // 10), replace=False)
    live = np.ones(n, bool)
    for b in blind:
        live[6 * b:6 * b + 6] = False
    m = int(live.sum())
    B = rng.standard_normal((m, 96))
    Hl = B @ B.T / 96 + np.diag(rng.uniform(0.5, 50.0, m))
    H = np.zeros((n, n)); H[np.ix_(live, live)] = Hl
    g = np.zeros(n); g[live] = rng.standard_normal(m)
    u = 0.1
    ref = np.zeros(n)
    ref[live] = np.linalg.solve(Hl + u * np.diag(np.diag(Hl)), -g[live])
    c = capi.Context(W)
    try:
        for mode in (None, "chainb", "chain", "fused", "launches"):
            if mode:
                os.environ["BALM_SOLVE"] = mode
            else:
                os.environ.pop("BALM_SOLVE", None)
            dx, q1 = c.solve_damped(H, g, u)
            assert np.all(dx[~live] == 0.0), mode
            assert rel_err(dx[live], ref[live]) < 1e-9, (mode, rel_err(dx[live], ref[live]))
    finally:
        os.environ.pop("BALM_SOLVE", None)
    c.close()


@pytest.mark.parametrize("W", [8, 9, 16, 17, 24, 33, 48, 64, 100, 144, 177, 200, 256, 320, 400, 500])
@pytest.mark.parametrize("kind", ["spd", "indefinite"])
def test_chain_kernel_matches_lapack(W, kind):
    """k_ldl_chain (round 3: one workgroup owns the diagonal, everybody else's rows are one product with Minv = L11^-T D11^-1)
    against LAPACK and against the launch path, solve after solve (flags are re-zeroed per solve)"""
    H, g = _test_matrix(W, kind, 77 * W + (kind == "spd"))
    u = 0.1
    ref = np.linalg.solve(H + u * np.diag(np.diag(H)), -g)
    c = capi.Context(W)
    os.environ["BALM_SOLVE"] = "chain"
    try:
        for _ in range(3):
            dx, q1 = c.solve_damped(H, g, u)
            assert np.all(np.isfinite(dx)), "k_ldl_chain gave up on a flag (bounded waits) or was not launched"
            assert rel_err(dx, ref) < 1e-9
        os.environ["BALM_SOLVE"] = "launches"
        dx0, q0 = c.solve_damped(H, g, u)
        assert rel_err(dx, dx0) < 1e-10 and abs(q1 - q0) <= 1e-10 * abs(q0)
    finally:
        os.environ.pop("BALM_SOLVE", None)
    c.close()


def test_chain_kernel_lm_run_and_real_hessian():
    sc, _ = make_scene(23, 64, 600, 8, drop=0.3)
    c = capi.Context(sc.W)
    c.set_features(sc.clusters, None, sc.coeffs)
    os.environ["BALM_SOLVE"] = "launches"
    try:
        pa, la = c.damping_iter(sc.poses_init, u0=0.01, max_iter=10)
        os.environ["BALM_SOLVE"] = "chain"
        pb, lb = c.damping_iter(sc.poses_init, u0=0.01, max_iter=10)
    finally:
        os.environ.pop("BALM_SOLVE", None)
    assert len(la) == len(lb) and np.allclose(la[:, :3], lb[:, :3], rtol=1e-9, atol=0) and np.abs(pa - pb).max() < 1e-10
    c.close()


# ---- k_solve_small (round 4): the whole damped solve of a window of <= 32 poses as one launch of one workgroup ----------------
@pytest.mark.parametrize("W", [1, 2, 5, 8, 9, 15, 16, 17, 20, 23, 24, 25, 31, 32])
@pytest.mark.parametrize("kind", ["spd", "indefinite"])
def test_small_window_kernel_matches_lapack(W, kind):
    """k_solve_small (rank + build + LDL^T + substitution + q1 in one launch, the matrix in LDS; bavoxel.hpp:1113-1127) against LAPACK and
    against the launch path, every panel count 1..3 and windows that do / do not fill their last panel, solve after solve; windows of
    25..32 poses (four panels) are beyond it and stay on the launch path"""
    H, g = _test_matrix(W, kind, 31 * W + (kind == "spd"))
    u = 0.1
    D = np.diag(np.diag(H))
    ref = np.linalg.solve(H + u * D, -g)
    c = capi.Context(W)
    try:
        os.environ["BALM_SOLVE"] = "small"
        for _ in range(3):
            dx, q1 = c.solve_damped(H, g, u)
            assert np.all(np.isfinite(dx))
            assert rel_err(dx, ref) < 1e-9
        assert abs(q1 - 0.5 * dx @ (u * D @ dx - g)) <= 1e-12 * abs(q1)
        os.environ["BALM_SOLVE"] = "launches"
        dx0, q0 = c.solve_damped(H, g, u)
        assert rel_err(dx, dx0) < 1e-10 and abs(q1 - q0) <= 1e-10 * abs(q0)
        os.environ.pop("BALM_SOLVE")
        dxd, qd = c.solve_damped(H, g, u)                    # the default IS the small kernel at these sizes: bit for bit
        assert np.array_equal(dxd, dx) and qd == q1
    finally:
        os.environ.pop("BALM_SOLVE", None)
    c.close()


@pytest.mark.parametrize("W", [12, 20, 30])
def test_small_window_kernel_blind_poses(W):
    """zero 6x6 blocks (poses that observe nothing): zero pivots, D^+ leaves exact zeros there (Eigen's rule), as on every other path"""
    rng = np.random.default_rng(W)
    n = 6 * W
    blind = rng.choice(W, size=3, replace=False)
    live = np.ones(n, bool)
    for b in blind:
        live[6 * b:6 * b + 6] = False
    m = int(live.sum())
    B = rng.standard_normal((m, 96))
    Hl = B @ B.T / 96 + np.diag(rng.uniform(0.5, 50.0, m))
    H = np.zeros((n, n)); H[np.ix_(live, live)] = Hl
    g = np.zeros(n); g[live] = rng.standard_normal(m)
    u = 0.1
    ref = np.zeros(n)
    ref[live] = np.linalg.solve(Hl + u * np.diag(np.diag(Hl)), -g[live])
    c = capi.Context(W)
    try:
        for mode in ("small", "launches"):
            os.environ["BALM_SOLVE"] = mode
            dx, q1 = c.solve_damped(H, g, u)
            assert np.all(dx[~live] == 0.0), mode
            assert rel_err(dx[live], ref[live]) < 1e-9, (mode, rel_err(dx[live], ref[live]))
    finally:
        os.environ.pop("BALM_SOLVE", None)
    c.close()


@pytest.mark.parametrize("W,F,form", [(20, 20, 0), (20, 150, 1), (32, 300, 0), (9, 40, 0)])
def test_small_window_kernel_lm_run(W, F, form):
    """an LM run with the one-launch solve (trial poses written by the kernel itself) against the launch path: same accept / reject
    sequence, same poses; plain launches and the replayed hipGraph"""
    sc, _ = make_scene(40 + W, W, F, 8, drop=0.2)
    c = capi.Context(sc.W)
    c.set_features(sc.clusters, None, sc.coeffs)
    try:
        os.environ["BALM_SOLVE"] = "launches"
        pa, la = c.damping_iter(sc.poses_init, form=form, u0=0.01, max_iter=10)
        os.environ.pop("BALM_SOLVE")
        pb, lb = c.damping_iter(sc.poses_init, form=form, u0=0.01, max_iter=10)
        os.environ["BALM_GRAPH"] = "1"
        pc, lc = c.damping_iter(sc.poses_init, form=form, u0=0.01, max_iter=10)
    finally:
        os.environ.pop("BALM_SOLVE", None); os.environ.pop("BALM_GRAPH", None)
    assert len(la) == len(lb) and np.allclose(la[:, :3], lb[:, :3], rtol=1e-9, atol=0) and np.abs(pa - pb).max() < 1e-10
    assert len(lc) == len(lb) and np.allclose(lc[:, :3], lb[:, :3], rtol=1e-12, atol=0) and np.abs(pc - pb).max() < 1e-12
    c.close()


@pytest.mark.parametrize("W", [8, 20, 40, 200])
def test_a_nan_in_the_hessian_neither_hangs_nor_sticks(W):
    """A NaN on the diagonal (it takes part in the pivot ranking) or off it must come back as a non-finite step -- not as a hang, a crash or a
    plausible-looking solution -- on the one-launch small-window kernel (W = 8, 20), the persistent chain (40, 200); and the context must
    solve a clean system correctly right afterwards (Eigen's ldlt().solve() likewise just propagates the NaN: bavoxel.hpp:1113-1114)."""
    H, g = _test_matrix(W, "spd", 5 * W)
    u = 0.1
    ref = np.linalg.solve(H + u * np.diag(np.diag(H)), -g)
    c = capi.Context(W)
    for where in ("diag", "off"):
        Hb = H.copy()
        if where == "diag":
            Hb[7, 7] = np.nan
        else:
            Hb[5, 11] = Hb[11, 5] = np.nan
        dx, q1 = c.solve_damped(Hb, g, u)
        assert not np.all(np.isfinite(dx)), where
        dx, q1 = c.solve_damped(H, g, u)
        assert rel_err(dx, ref) < 1e-9, where
    c.close()


def test_lm_loop_reports_non_finite_input_on_a_small_window():
    """NaN poses into balm_damping_iter on a window that takes the one-launch solve: BALM_ERR_NUMERIC, and the context stays usable"""
    sc, _ = make_scene(77, 12, 80, 8, drop=0.1)
    c = capi.Context(sc.W)
    c.set_features(sc.clusters, None, sc.coeffs)
    bad = sc.poses_init.copy()
    bad[3, 9] = np.nan
    with pytest.raises(capi.BalmError) as e:
        c.damping_iter(bad, u0=0.01, max_iter=5)
    assert e.value.code == capi.ERR_NUMERIC
    out, lg = c.damping_iter(sc.poses_init, u0=0.01, max_iter=5)
    assert np.all(np.isfinite(out)) and lg[-1, 1] < lg[0, 0]
    c.close()


def test_solve_trace_switch_records_the_chain_phases(monkeypatch):
    """BALM_SOLVE_TRACE=1 at balm_create: balm_get_solve_trace returns the persistent kernel's per-(row block, column, phase) ticks
    of the last factorisation (tools/chain_check.py reads them); without the switch: no buffer, BALM_ERR_STATE."""
    monkeypatch.setenv("BALM_SOLVE_TRACE", "1")
    W = 100
    n = 6 * W
    rng = np.random.default_rng(3)
    B = rng.standard_normal((n, n))
    H = B @ B.T / n + np.diag(rng.uniform(0.5, 5.0, n))
    g = rng.standard_normal(n)
    c = capi.Context(W)
    dx, _ = c.solve_damped(H, g, 0.1)
    assert rel_err(dx, np.linalg.solve(H + 0.1 * np.diag(np.diag(H)), -g)) < 1e-9
    tr = c.solve_trace()
    assert tr.shape[2] == 6 and (tr != 0).any()
    c.close()
    monkeypatch.delenv("BALM_SOLVE_TRACE")
    c = capi.Context(W)
    c.solve_damped(H, g, 0.1)
    with pytest.raises(capi.BalmError):
        c.solve_trace()
    c.close()


@pytest.mark.parametrize("W,F", [(100, 400), (200, 900), (40, 200)])
def test_a_timed_out_persistent_solve_is_retried_on_the_launch_path(W, F, monkeypatch):
    """A wait inside k_ldl_chain that hits its poll limit (its workgroups not all resident: another process or stream holds CUs)
    raises the abort flag and k_ldl_finish poisons the step.  The LM loop then repeats that iteration's solve on the launch path and
    stays there (BALM_FAULT_INJECT="timeout,<iteration>" raises the flag the way the kernel would): same accept sequence, same poses
    to rounding as the undisturbed run, no BALM_ERR_NUMERIC."""
    sc, _ = make_scene(5, W, F, 6, drop=0.2)
    c = capi.Context(W)
    c.set_features(sc.clusters, None, sc.coeffs)
    p0, l0 = c.damping_iter(sc.poses_init, u0=0.01, max_iter=10)
    c.close()
    monkeypatch.setenv("BALM_FAULT_INJECT", "timeout,1")
    monkeypatch.setenv("BALM_SOLVE_DEBUG", "1")
    c = capi.Context(W)
    c.set_features(sc.clusters, None, sc.coeffs)
    p1, l1 = c.damping_iter(sc.poses_init, u0=0.01, max_iter=10)
    assert len(l0) == len(l1) and np.array_equal(l0[:, 6], l1[:, 6])
    assert np.allclose(l0[:, :3], l1[:, :3], rtol=1e-9, atol=0) and np.abs(p0 - p1).max() < 1e-10
    # ... and the context keeps solving (on the launch path) afterwards
    monkeypatch.delenv("BALM_FAULT_INJECT")
    p2, l2 = c.damping_iter(sc.poses_init, u0=0.01, max_iter=10)
    assert np.abs(p2 - p0).max() < 1e-10
    c.close()


def test_two_live_contexts_on_one_device_solve_at_the_same_time():
    """Plain launches of the persistent kernels need every workgroup resident; two ordinary contexts driven from two threads
    each size their grids for HALF the device's slots (contexts alive per device are counted), so neither can starve the other
    into its poll limit.  40 concurrent solves per thread at n = 1200, each against LAPACK."""
    import threading
    W = 200
    n = 6 * W
    rng = np.random.default_rng(11)
    B = rng.standard_normal((n, n))
    H = B @ B.T / n + np.diag(rng.uniform(0.5, 5.0, n))
    g = rng.standard_normal(n)
    ref = np.linalg.solve(H + 0.1 * np.diag(np.diag(H)), -g)
    ctxs = [capi.Context(W), capi.Context(W)]
    errs = [[], []]

    def work(k):
        for _ in range(40):
            dx, _ = ctxs[k].solve_damped(H, g, 0.1)
            errs[k].append(rel_err(dx, ref))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert len(errs[0]) == 40 and len(errs[1]) == 40 and max(errs[0] + errs[1]) < 1e-9
    for c in ctxs:
        c.close()
