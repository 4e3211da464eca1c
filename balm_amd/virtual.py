"""ROS-free counterpart of the reference's benchmark_virtual driver (src/benchmark/benchmark_virtual.cpp:523-600)
for the method this package implements: generate a window of poses and plane patches, perturb the poses, build
the point clusters on the device, run BALM2's LM loop (that file's constants: u0 = 0.1, <= 20 iterations, weight
winSize * ptsSize) and print what the reference prints.

    python -m balm_amd.virtual [--winSize 20] [--sufSize 150] [--ptsSize 40] [--point_noise 0.05] [--surf_range 2.0]
"""
import sys
import time

import numpy as np

from . import capi, scene


def _log_so3(R):
    c = min(1.0, max(-1.0, (np.trace(R) - 1.0) / 2.0))
    th = np.arccos(c)
    k = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return 0.5 * k if th < 1e-9 else 0.5 * th / np.sin(th) * k


def rsme(poses_gt, poses_es):
    """benchmark_virtual.cpp:48-61 (after re-anchoring both to pose 0, :472-479)"""
    def anchored(P):
        R = P[:, :9].reshape(-1, 3, 3).transpose(0, 2, 1)
        p = P[:, 9:]
        return np.einsum("ji,njk->nik", R[0], R), (p - p[0]) @ R[0]
    Rg, pg = anchored(poses_gt)
    Re, pe = anchored(poses_es)
    rot = np.sqrt(np.mean([np.sum(_log_so3(Rg[i].T @ Re[i]) ** 2) for i in range(Rg.shape[0])]))
    tran = np.sqrt(np.mean(np.sum((pe - pg) ** 2, axis=1)))
    return rot, tran


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--winSize", type=int, default=20)            # launch/benchmark_virtual.launch defaults
    ap.add_argument("--sufSize", type=int, default=150)
    ap.add_argument("--ptsSize", type=int, default=40)
    ap.add_argument("--point_noise", type=float, default=0.05)
    ap.add_argument("--surf_range", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    print("winSize: %d\nsufSize: %d\npstSize: %d" % (a.winSize, a.sufSize, a.ptsSize), flush=True)   # :541-543 (sic)
    seed = int(time.time()) if a.seed is None else a.seed                                 # :545
    sc = scene.generate(seed, a.winSize, a.sufSize, a.ptsSize, point_noise=a.point_noise, surf_range=a.surf_range,
                        keep_points=True)
    W, F, P = a.winSize, a.sufSize, a.ptsSize
    ctx = capi.Context(W, a.device)
    xyz = sc.points.reshape(-1, 3)
    fid = np.repeat(np.arange(F, dtype=np.int32), W * P)
    pid = np.tile(np.repeat(np.arange(W, dtype=np.int32), P), F)
    t0 = time.time()
    ctx.build_clusters(F, xyz, fid, pid, None, np.full(F, float(W * P)), want_clusters=False)      # :391-403
    out, lg = ctx.damping_iter(sc.poses_init, form=capi.FORM_LEFT, u0=0.1, max_iter=20, verbose=True)   # :380, :408
    dt = time.time() - t0
    rot, tran = rsme(sc.poses_gt, out)
    sys.stdout.flush()
    print("RSME: %fdeg, %fm" % (rot * 57.3, tran))                                           # :518
    print("(%d LM iterations, %.1f ms on the GPU including the cluster build)" % (len(lg), dt * 1e3))
    return 0


if __name__ == "__main__":
    sys.exit(main())
