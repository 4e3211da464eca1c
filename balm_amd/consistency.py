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
