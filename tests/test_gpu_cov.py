"""N4: point-noise -> pose covariance on the GPU (balm_pose_covariance, csrc/kernels_cov.hip)
against golden vectors made by the reference's own compiled sources, the numpy oracle, the compiled reference
where it travels, and the domain's own acceptance test: the NEES of Monte-Carlo runs is 6 W
(src/simulation/consistency.cpp:168-170).  Tolerances relative to the largest entry: 1e-10 for Rcov_raw
(FP64 sums in a different order + a 3x3 Cholesky per feature), 1e-8 for Rcov (two solves with cond(H) ~ 1e3)."""
import os

import numpy as np
import pytest

from balm_amd import capi, consistency, scene
from conftest import ROOT
from oracle import numpy_oracle as npo
from oracle import ref_sim

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def test_pose_covariance_matches_reference_golden():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "cov_w6_f10.npz")))
    c = capi.Context(6)
    c.set_features(g["clusters"], g["fix"], np.ones(10))
    Rcov, Rraw = c.pose_covariance(g["poses"], cluster_cov=g["ccov"])
    assert rel(Rraw, g["Rraw"]) < 1e-10 and rel(Rcov, g["Rcov"]) < 1e-8
    assert np.array_equal(Rcov, Rcov.T) or rel(Rcov, Rcov.T) < 1e-12
    # isotropic point noise rebuilt on the device from the clusters = what PointCluster::push accumulated
    Rcov2, Rraw2 = c.pose_covariance(g["poses"], point_sigma=float(g["pn"]))
    assert rel(Rraw2, g["Rraw"]) < 1e-10 and rel(Rcov2, g["Rcov"]) < 1e-8
    # run-to-run bit-identical (no atomics)
    Rcov3, Rraw3 = c.pose_covariance(g["poses"], point_sigma=float(g["pn"]))
    assert np.array_equal(Rraw2, Rraw3) and np.array_equal(Rcov2, Rcov3)
    c.close()


def anchored_scene(seed, W, F, pts, sparse=False):
    """W free poses + the scan of an extra, marginalised pose as fix clusters (consistency.cpp's win + fix window)"""
    sc = scene.generate(seed, W + 1, F, pts, keep_points=True)
    cl = sc.clusters[:, 1:].copy()
    if sparse:
        rng = np.random.default_rng(seed)
        cl[rng.random((F, W)) < 0.5] = 0
        cl[:, 0] = sc.clusters[:, 1]                        # every feature keeps one observer
    pts0 = sc.points[:, 0].astype(np.float64)               # [F, pts, 3] body frame of pose 0
    R0, p0 = npo.pose_R(sc.poses_gt)[0], npo.pose_p(sc.poses_gt)[0]
    w = pts0 @ R0.T + p0
    fix = np.zeros((F, 10))
    fix[:, 0] = (w[..., 0] ** 2).sum(1); fix[:, 1] = (w[..., 0] * w[..., 1]).sum(1); fix[:, 2] = (w[..., 0] * w[..., 2]).sum(1)
    fix[:, 3] = (w[..., 1] ** 2).sum(1); fix[:, 4] = (w[..., 1] * w[..., 2]).sum(1); fix[:, 5] = (w[..., 2] ** 2).sum(1)
    fix[:, 6:9] = w.sum(1); fix[:, 9] = w.shape[1]
    return cl, fix, sc.poses_init[1:].copy(), sc.poses_gt[1:].copy(), sc


@pytest.mark.parametrize("seed,W,F,sparse", [(1, 20, 60, False), (2, 33, 100, True), (3, 64, 150, True), (4, 5, 3, False)])
def test_pose_covariance_matches_oracle(seed, W, F, sparse):
    cl, fix, poses, _, _ = anchored_scene(seed, W, F, 20, sparse)
    co = np.linspace(0.7, 1.6, F)
    cc = npo.cluster_noise_cov_closed_form(cl, 0.05) * (1.0 + 0.1 * np.arange(W))[None, :, None, None]   # not isotropic-sigma
    Rf = npo.point_cov_left_factored(cl, cc, fix, poses, coeffs=co)[0]
    H, _, _ = npo.left_evaluate(cl, fix, co, poses)
    c = capi.Context(W)
    c.set_features(cl, fix, co)
    Rcov, Rraw = c.pose_covariance(poses, cluster_cov=cc)
    assert rel(Rraw, Rf) < 1e-10
    assert rel(Rcov, npo.pose_cov(H, Rf)) < 1e-8
    c.close()


def test_pose_covariance_matches_compiled_reference():
    if not ref_sim.available():
        pytest.skip("oracle/_ref/libbalm_ref_sim.so not built (needs /root/reference at build time)")
    cl, fix, poses, _, _ = anchored_scene(7, 12, 25, 25, sparse=True)
    cc = npo.cluster_noise_cov_closed_form(cl, 0.02)
    Hr, Rc = ref_sim.pose_cov(cl, cc, fix, poses)
    Rr = ref_sim.point_cov(cl, cc, fix, poses)
    c = capi.Context(12)
    c.set_features(cl, fix, np.ones(25))
    Rcov, Rraw = c.pose_covariance(poses, point_sigma=0.02)
    assert rel(Rraw, Rr) < 1e-10 and rel(Rcov, Rc) < 1e-8
    c.close()


def test_nees_of_monte_carlo_runs_is_6w():
    """the consistency experiment: corrupt the points with N(0, pn^2) noise, optimise, compare the error against
    ground truth with the predicted covariance.  NEES ~ chi^2(6W): mean 6W, std sqrt(12W)."""
    W, F, PTS, PN, RUNS = 10, 120, 40, 0.02, 12
    vals = []
    for run in range(RUNS):
        sc = scene.generate(100 + run, W + 1, F, PTS, point_noise=0.0, keep_points=True)      # noise-free geometry
        rng = np.random.default_rng(1000 + run)
        pts = sc.points.astype(np.float64)                                               # [F, W+1, PTS, 3]
        noisy = pts[:, 1:] + PN * rng.standard_normal(pts[:, 1:].shape)
        c = capi.Context(W)
        xyz = noisy.reshape(-1, 3).astype(np.float32)
        fid = np.repeat(np.arange(F), W * PTS).astype(np.int32)
        pid = np.tile(np.repeat(np.arange(W), PTS), F).astype(np.int32)
        R0, p0 = npo.pose_R(sc.poses_gt)[0], npo.pose_p(sc.poses_gt)[0]
        w = pts[:, 0] @ R0.T + p0                                                        # the marginalised scan: exact
        fix = np.zeros((F, 10))
        fix[:, 0] = (w[..., 0] ** 2).sum(1); fix[:, 1] = (w[..., 0] * w[..., 1]).sum(1); fix[:, 2] = (w[..., 0] * w[..., 2]).sum(1)
        fix[:, 3] = (w[..., 1] ** 2).sum(1); fix[:, 4] = (w[..., 1] * w[..., 2]).sum(1); fix[:, 5] = (w[..., 2] ** 2).sum(1)
        fix[:, 6:9] = w.sum(1); fix[:, 9] = PTS
        c.build_clusters(F, xyz, fid, pid, fix, np.ones(F), want_clusters=False)
        gt = sc.poses_gt[1:]
        est, lg = c.damping_iter(gt.copy(), form=0, u0=0.01, max_iter=30, rel_tol=1e-12, reanchor=False)
        Rcov, _ = c.pose_covariance(est, point_sigma=PN, want_raw=False)
        v, err = consistency.nees(c, est, gt, Rcov)
        vals.append(v)
        c.close()
    vals = np.array(vals)
    print("NEES over %d runs: mean %.1f (expected %d), min %.1f max %.1f" % (RUNS, vals.mean(), 6 * W, vals.min(), vals.max()))
    # mean of RUNS chi^2(6W) variables: 6W +- sqrt(12W / RUNS); accept +-5 sigma, single runs +-6 sigma
    assert abs(vals.mean() - 6 * W) < 5 * np.sqrt(12 * W / RUNS)
    assert np.all(np.abs(vals - 6 * W) < 6 * np.sqrt(12 * W))


def test_pose_covariance_errors_and_timing():
    c = capi.Context(8)
    with pytest.raises(capi.BalmError):
        c.pose_covariance(np.zeros((8, 12)), point_sigma=0.02)            # no features
    cl, fix, poses, _, _ = anchored_scene(5, 8, 12, 15)
    c.set_features(cl, fix, np.ones(12))
    with pytest.raises(capi.BalmError):
        c.pose_covariance(poses)                                           # neither covariances nor a sigma
    c.close()
    # the consistency experiment's size (win 100) and the bench window
    for W, F in ((100, 2000), (200, 20000)):
        sc = scene.generate(9, W, F, 6)
        fix = 0.3 * sc.clusters[:, 0]
        fix[:, 9] = np.round(fix[:, 9])
        c = capi.Context(W, flags=capi.FLAG_TIMING)
        c.set_features(sc.clusters, fix, np.ones(F))
        est, _ = c.damping_iter(sc.poses_init, form=0, u0=0.01, max_iter=20, reanchor=False)   # the experiment's flow: at the optimum
        c.pose_covariance(est, point_sigma=0.02, want_raw=False)
        c.reset_timing()
        Rcov, _ = c.pose_covariance(est, point_sigma=0.02, want_raw=False)
        t = c.timing()
        assert np.isfinite(Rcov).all() and np.abs(Rcov - Rcov.T).max() <= 1e-11 * np.abs(Rcov).max()
        assert np.linalg.eigvalsh(Rcov).min() > 0                        # a covariance
        print("pose covariance W=%d F=%d: covariance stage %.2f ms (factors, 2 SYRKs, LDL, 4 products with L^-T D^+) + Hessian %.2f ms"
              % (W, F, t["cov"][0], t["moments"][0] + t["factors"][0] + t["syrk"][0] + t["assemble"][0]))
        c.close()


def test_cpp_shim_dropin_for_the_consistency_driver():
    """include/balm_shim.hpp against the consistency driver's call (consistency.cpp:150-156): the reference's own
    PointCluster::push (with c_cov) fills a VOX_HESS; BALM2::damping_iter(x, voxhess, Rcov) (CPU, compiled from
    /root/reference/src/simulation) and BALM2_HIP's run on the same container -- tests/cpp/shim_sim_driver.cpp."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "shim_sim_driver")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/shim_sim_driver not built (needs /root/reference at build time)")
    scene_so = os.path.join(ROOT, "balm_amd", "lib", "libbalm_scene.so")
    p = subprocess.run([exe, "3", "8", "40", "30", scene_so], cwd=ROOT, capture_output=True, text=True, timeout=900)
    line = [l for l in p.stdout.splitlines() if l.startswith("SHIM_SIM_DRIVER")]
    assert line, p.stdout[-2000:] + p.stderr[-2000:]
    print(line[-1])
    assert p.returncode == 0, line[-1]


def test_consistency_experiment_on_shipped_scans():
    """src/simulation/consistency.cpp end to end on its own shipped data (datas/consistency: 101 simulated scans):
    association with that driver's rules (on the device; once more on the host), per run noise -> device cluster build -> device LM -> device
    covariance -> NEES.  The reference prints "The expected NEES is 6*100 = 600"."""
    path = os.path.join(ROOT, "oracle", "_ref", "consistency_scans.npz")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/consistency_scans.npz not built (needs /root/reference/datas)")
    d = np.load(path)
    frames = np.split(d["xyz"], np.cumsum(d["counts"])[:-1])
    c = capi.Context(100)
    vals, F = consistency.monte_carlo(c, frames, d["poses"], pnoise=0.02, runs=4, seed=7)          # association on the GPU too
    from oracle import assoc_host as ah
    from balm_amd import realworld as rw
    vals_h, F_h = consistency.monte_carlo(c, frames, d["poses"], pnoise=0.02, runs=1, seed=7,
                                          association=ah.associate(frames, d["poses"], want_points=True, **rw.SIM_RULES))
    assert F_h == F and abs(vals_h[0] - 600) < 6 * np.sqrt(1200)         # same features; the noise lands on the points in another order
    c.close()
    vals = np.array(vals)
    print("consistency experiment: %d features, NEES %s (expected 600 +- 35)" % (F, np.round(vals, 1)))
    assert F == 1096
    assert np.all(np.abs(vals - 600) < 6 * np.sqrt(1200)) and abs(vals.mean() - 600) < 5 * np.sqrt(1200 / 4)


def test_pose_covariance_wide_window():
    """W > 256: every lane of the per-feature workgroups owns more than one pose; n = 1800 -> 38 LDL panels"""
    W, F = 300, 14
    cl, fix, poses, _, _ = anchored_scene(21, W, F, 8, sparse=True)
    cc = npo.cluster_noise_cov_closed_form(cl, 0.03)
    Rf = npo.point_cov_left_factored(cl, cc, fix, poses)[0]
    H, _, _ = npo.left_evaluate(cl, fix, np.ones(F), poses)
    c = capi.Context(W)
    c.set_features(cl, fix, np.ones(F))
    Rcov, Rraw = c.pose_covariance(poses, point_sigma=0.03)
    assert rel(Rraw, Rf) < 1e-10
    # few features over many poses: H is poorly conditioned, compare through the better-posed product H Rcov H^T
    assert rel(H @ Rcov @ H.T, Rf) < 1e-7
    c.close()


def _cov_two_rank_worker(rank, world, port, seed, W, F, q):
    import torch.distributed as dist
    from balm_amd import dist as bdist
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    bdist.init_process_group("gloo")          # two ranks on the box's one GPU: RCCL refuses that, gloo does not
    cl, fix, poses, _, _ = anchored_scene(seed, W, F, 12, sparse=True)
    nobs = (cl[..., 9] > 0).sum(1)
    lo, hi = bdist.partition_features(nobs, world)[rank]
    c = capi.Context(W, 0)
    c.set_features(cl[lo:hi], fix[lo:hi], np.ones(hi - lo))
    bdist.install_allreduce(c)
    Rcov, Rraw = c.pose_covariance(poses, point_sigma=0.02)
    if rank == 0:
        q.put((Rcov, Rraw))
    dist.barrier()
    c.close()
    dist.destroy_process_group()


def test_pose_covariance_two_ranks_sharded_on_one_gpu():
    """the N>1 path of the covariance stage: feature shards on two processes, [XX^T tiles | YY^T tiles | S] summed
    through the balm_set_allreduce hook, the solves replicated; equals the single-process result"""
    import socket
    import torch.multiprocessing as mp
    seed, W, F = 31, 16, 70
    cl, fix, poses, _, _ = anchored_scene(seed, W, F, 12, sparse=True)
    c = capi.Context(W)
    c.set_features(cl, fix, np.ones(F))
    Rcov1, Rraw1 = c.pose_covariance(poses, point_sigma=0.02)
    c.close()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cov_two_rank_worker, args=(r, 2, port, seed, W, F, q)) for r in range(2)]
    for p in procs:
        p.start()
    Rcov2, Rraw2 = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert rel(Rraw2, Rraw1) < 1e-12 and rel(Rcov2, Rcov1) < 1e-9


def test_consistency_experiment_incremental_association():
    """the same experiment with the driver's OWN association sequence (consistency.cpp:108-136: cut_voxel for 101 scans, one
    recut, one marginalize) on the device map (balm_window_* with the strict plane test, fix_frames, defer_recut) and the
    feature points read back from the map: the feature set of the batch association bit for bit, the points rebuild the
    clusters, NEES about 600"""
    path = os.path.join(ROOT, "oracle", "_ref", "consistency_scans.npz")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/consistency_scans.npz not built (needs /root/reference/datas)")
    from balm_amd import realworld as rw
    d = np.load(path)
    frames = np.split(d["xyz"], np.cumsum(d["counts"])[:-1])
    c = capi.Context(100)
    cl, co, layer, fix, (xyz, fid, sid) = consistency.associate_incremental(c, frames, d["poses"])
    assert cl.shape[0] == 1096
    Fb, _, (clb, cob, layb, fixb, _) = rw.associate_gpu(c, frames, d["poses"], want_points=True, **rw.SIM_RULES)
    key = lambda a, f: np.concatenate([a.reshape(a.shape[0], -1), f], axis=1)
    ka, kb = key(cl, fix), key(clb, fixb)
    assert np.array_equal(ka[np.lexsort(ka[:, ::-1].T)], kb[np.lexsort(kb[:, ::-1].T)])
    # the points the map hands out are the points of the clusters: counts per (feature, scan)
    N = np.zeros(cl.shape[:2])
    np.add.at(N, (fid, sid), 1)
    assert np.array_equal(N, cl[:, :, 9])
    vals, F = consistency.monte_carlo(c, frames, d["poses"], pnoise=0.02, runs=2, seed=11, association=(cl, co, layer, fix, (xyz, fid, sid)))
    c.close()
    print("consistency experiment, incremental association: %d features, NEES %s" % (F, np.round(vals, 1)))
    assert np.all(np.abs(np.array(vals) - 600) < 6 * np.sqrt(1200))


@pytest.mark.parametrize("seed,W,F,sparse,explicit", [(5, 20, 60, False, True), (6, 100, 90, True, False), (7, 200, 64, True, False), (8, 70, 40, False, True)])
def test_cov_factor_kernel_paths_agree(seed, W, F, sparse, explicit, monkeypatch):
    """k_cov_factors keeps a pose's At / Rr rows in registers across the block-wide sum of Q and writes the X / Y columns once, coalesced
    (windows of <= 256 poses, the default) or parks them in the columns and re-reads them (BALM_COV_ONEPASS=0, wider windows): same
    arithmetic in the same order (BAs_left.hpp:418-450 is what both evaluate) -> the same covariance bit for bit; one, two and four
    wavefronts of poses, explicit cluster covariances and the isotropic closed form"""
    cl, fix, poses, _, _ = anchored_scene(seed, W, F, 12, sparse)
    cc = npo.cluster_noise_cov_closed_form(cl, 0.05) if explicit else None
    out = []
    for mode in ("0", "1"):
        monkeypatch.setenv("BALM_COV_ONEPASS", mode)
        c = capi.Context(W)
        c.set_features(cl, fix, np.ones(F))
        out.append(c.pose_covariance(poses, cluster_cov=cc) if explicit else c.pose_covariance(poses, point_sigma=0.05))
        c.close()
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])
