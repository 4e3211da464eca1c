"""Host side of the consistency experiment ("next" row N4): what src/simulation/consistency.cpp does around the
optimizer -- the left-invariant pose error against ground truth and the NEES of one Monte-Carlo run -- on top
of balm_pose_covariance / balm_solve_damped.  Only glue here: the arithmetic that matters runs on the GPU."""
import numpy as np


def _log_so3(R):
    c = min(1.0, max(-1.0, (np.trace(R) - 1.0) / 2.0))
    th = np.arccos(c)
    k = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return 0.5 * k if th < 1e-9 else 0.5 * th / np.sin(th) * k


def pose_error_left(poses_est, poses_gt):
    """consistency.cpp:159-166: err_i = [Log(R_gt R_est^T) ; -R_gt R_est^T p_est + p_gt]  (poses [W,12])"""
    W = poses_est.shape[0]
    Re = poses_est[:, :9].reshape(W, 3, 3).transpose(0, 2, 1)
    Rg = poses_gt[:, :9].reshape(W, 3, 3).transpose(0, 2, 1)
    err = np.zeros(6 * W)
    for i in range(W):
        dR = Rg[i] @ Re[i].T
        err[6 * i:6 * i + 3] = _log_so3(dR)
        err[6 * i + 3:6 * i + 6] = -dR @ poses_est[i, 9:] + poses_gt[i, 9:]
    return err


def nees(ctx, poses_est, poses_gt, Rcov):
    """consistency.cpp:168: err^T Rcov^-1 err, the solve on the device (expected value 6 W)"""
    err = pose_error_left(poses_est, poses_gt)
    _, q1 = ctx.solve_damped(Rcov, -err, 0.0)       # q1 = 0.5 x.(0 - (-err)) with Rcov x = err
    return 2.0 * q1, err


def load_window(data_dir, n_poses=101):
    """consistency.cpp:57-94: lidarPose.csv (4 text lines per pose) + <m>.pcd, m = 1..n; translations relative to
    pose 0 (rotations are kept, :87-88)"""
    import os
    from . import realworld as rw
    poses, _ = rw.read_pose_csv(os.path.join(data_dir, "lidarPose.csv"), n_poses)
    poses = poses.copy()
    poses[:, 9:] -= poses[0, 9:].copy()
    frames = [rw.read_pcd_xyz(os.path.join(data_dir, "%d.pcd" % (m + 1))) for m in range(poses.shape[0])]
    return poses, frames


def associate_incremental(ctx, frames, poses):
    """the driver's own sequence (consistency.cpp:108-136) on the device map: cut_voxel for win_size + fix_size scans, ONE
    recut, ONE marginalize(fix_size, {}, win_count), tras_opt -- and the points of the features for the noise runs.
    -> (clusters, coeffs, layer, fix, (xyz, feature, scan)) like the batch association's tuple"""
    from . import realworld as rw
    R = rw.SIM_RULES
    ctx.window_open(voxel_size=R["voxel_size"], eigen_thresholds=R["eigen_thresholds"], min_ps=R["min_ps"], layer_limit=R["layer_limit"],
                    min_observers=R["min_observers"], fix_frames=R["fix_frames"], strict=R["strict"], fix_point_limit=30,
                    defer_recut=True)
    for f, p in zip(frames, poses):
        ctx.window_add_scan(f, p)
    ctx.window_recut()
    ctx.window_marginalize(R["fix_frames"])
    F, (cl, co, layer, fix) = ctx.window_features()
    xyz, slot, feat = ctx.window_points()
    ctx.window_close()
    keep = feat >= 0
    return cl, co, layer, fix, (xyz[keep], feat[keep], slot[keep])


def monte_carlo(ctx, frames, poses, pnoise=0.02, runs=1, seed=0, verbose=False, association=None):
    """consistency.cpp:96-170 around the GPU path: associate the noise-free scans (first scan marginalised into
    fix clusters), then per run corrupt every feature point with N(0, pnoise^2) (OCTO_TREE_NODE::corrupt,
    BAs_left.hpp:886-906), rebuild the clusters on the device, optimise from the true poses, predict the covariance
    and score the error.  ctx: a capi.Context for len(frames) - 1 poses.  `association` (tests): a precomputed
    (clusters, coeffs, layer, fix, (xyz, feature, scan)) instead of the device association.  -> list of NEES, number of features"""
    from . import realworld as rw
    if association is None:
        F, _, (cl, co, layer, fix, pf) = rw.associate_gpu(ctx, frames, poses, want_points=True, **rw.SIM_RULES)
        allxyz = np.concatenate(frames)
        scan = np.concatenate([np.full(f.shape[0], i, np.int32) for i, f in enumerate(frames)]) - rw.SIM_RULES["fix_frames"]
        keep = (pf >= 0) & (scan >= 0)
        xyz, fid, sid = allxyz[keep], pf[keep], scan[keep]
    else:
        cl, co, layer, fix, (xyz, fid, sid) = association
    F, W = cl.shape[0], cl.shape[1]
    gt = poses[rw.SIM_RULES["fix_frames"]:]
    out = []
    for run in range(runs):
        rng = np.random.default_rng(seed + run)
        noisy = (xyz.astype(np.float64) + pnoise * rng.standard_normal(xyz.shape)).astype(np.float32)
        ctx.build_clusters(F, noisy, fid, sid, fix, np.ones(F), want_clusters=False)     # weight 1: BAs_left.hpp:44
        est, lg = ctx.damping_iter(gt, form=0, u0=0.01, max_iter=1000, rel_tol=0.0, abs_tol=1e-9, reanchor=False,
                                   verbose=verbose)
        Rcov, _ = ctx.pose_covariance(est, point_sigma=pnoise, want_raw=False)
        v, _ = nees(ctx, est, gt, Rcov)
        out.append(v)
    return out, F


def main(argv=None):
    import argparse
    import sys
    import time
    ap = argparse.ArgumentParser(description="the consistency experiment (src/simulation/consistency.cpp) on the GPU path")
    ap.add_argument("data_dir", help=".../datas/consistency")
    ap.add_argument("--pnoise", type=float, default=0.02)
    ap.add_argument("--runs", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--batch", action="store_true", help="associate with balm_associate (all scans at once) instead of the driver's own "
                                                         "incremental sequence on the device map")
    a = ap.parse_args(argv)
    from . import capi
    poses, frames = load_window(a.data_dir)
    W = poses.shape[0] - 1
    print("The size of poses: %d" % W)                                                  # consistency.cpp:138
    ctx = capi.Context(W, a.device)
    t = time.time()
    assoc = None if a.batch else associate_incremental(ctx, frames, poses)
    vals, F = monte_carlo(ctx, frames, poses, a.pnoise, a.runs, a.seed, verbose=a.runs == 1, association=assoc)
    print("%d plane features, %d run(s) in %.2f s" % (F, a.runs, time.time() - t))
    print("The expected NEES is 6*%d = %d." % (W, 6 * W))                              # :169
    for v in vals:
        print("The NEES for this Monto-Carlo experiment is %f." % v)                     # :170
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
