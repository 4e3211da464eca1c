"""Sliding-window bundle adjustment over a stream of scans (BASELINE configs[4]'s "sliding window"): the calling sequence the
reference's octree is built for -- cut_voxel into the live map and recut per scan (bavoxel.hpp:1170-1223, :737-776), tras_opt
(:908-929) + BALM2::damping_iter (:1069-1157, without the final re-anchor: the fix clusters pin the gauge, as in
src/simulation/BAs_left.hpp:1025-1100) once the window is full, OCTO_TREE_ROOT::marginalize with the optimised poses
(:948-963; consistency.cpp:127-136 does this once) -- with the map, the features and the optimiser on the device
(balm_window_* + balm_damping_iter).  Nothing but the scan upload and the pose download crosses the bus.

    python -m balm_amd.sliding /path/to/datas/benchmark_realworld --window 20 --slide 5 [--max-scans 60]
"""
import sys
import time

import numpy as np


def compose(a, b):
    """a o b for poses [12] = column-major R | p"""
    Ra, Rb = a[:9].reshape(3, 3).T, b[:9].reshape(3, 3).T
    out = np.empty(12)
    out[:9] = (Ra @ Rb).T.reshape(9)
    out[9:] = Ra @ b[9:] + a[9:]
    return out


def inverse(a):
    Ra = a[:9].reshape(3, 3).T
    out = np.empty(12)
    out[:9] = Ra.reshape(9)            # (R^T) column-major = R row-major
    out[9:] = -Ra.T @ a[9:]
    return out


class SlidingWindowBA:
    """ctx: a capi.Context for W poses.  push(scan, odometry pose) -> None, or the dict of a finished window."""

    def __init__(self, ctx, slide, voxel_size=2.0, eigen_thresholds=(1.0 / 16, 1.0 / 16, 1.0 / 9), min_ps=15, layer_limit=2,
                 u0=0.01, max_iter=10, min_planes=20, optimise=None):
        assert 1 <= slide <= ctx.W
        self.ctx, self.W, self.slide = ctx, ctx.W, slide
        self.lm = dict(u0=u0, max_iter=max_iter, min_planes=min_planes)
        self.optimise = optimise           # tests swap the optimiser (e.g. the reference's) in; default: the device LM loop
        ctx.window_open(voxel_size, eigen_thresholds, min_ps, layer_limit)
        self.odom, self.est = [], []       # poses of the scans in the window: as the odometry gave them / current estimate
        self.done = []                     # estimates of the scans that left the window
        self.windows = 0

    def push(self, xyz, pose_odom):
        pose_odom = np.asarray(pose_odom, dtype=np.float64).reshape(12)
        if self.est:                       # chain the odometry increment onto the refined pose of the previous scan
            guess = compose(self.est[-1], compose(inverse(self.odom[-1]), pose_odom))
        else:
            guess = pose_odom.copy()
        self.ctx.window_add_scan(xyz, guess)
        self.odom.append(pose_odom); self.est.append(guess)
        if len(self.est) < self.W:
            return None
        return self._optimise_and_slide()

    def _optimise_and_slide(self):
        ctx = self.ctx
        F, feats = ctx.window_features(want_features=self.optimise is not None)
        poses = np.stack(self.est)
        t0 = time.perf_counter()
        skipped = None
        if F == 0:
            # no plane survived the association: nothing to optimise, but the window must keep sliding
            out, log, skipped = poses.copy(), np.zeros((0, 8)), "no features"
        elif self.optimise is not None:
            out, log = self.optimise(feats, poses)
        else:
            from . import capi
            try:
                out, log = ctx.damping_iter(poses, form=0, reanchor=False, **self.lm)
            except capi.BalmError as e:
                # the reference prints its message and stops optimising this window (bavoxel.hpp:1079-1085, there followed by
                # exit(0)); a stream of scans goes on: keep the odometry-chained poses and slide
                if e.code != capi.ERR_TOO_FEW_PLANES:
                    raise
                out, log, skipped = poses.copy(), np.zeros((0, 8)), "a pose sees fewer than %d planes" % self.lm["min_planes"]
        dt = time.perf_counter() - t0
        ctx.window_marginalize(self.slide, out)
        self.done.extend(out[:self.slide])
        self.est = list(out[self.slide:])
        self.odom = self.odom[self.slide:]
        self.windows += 1
        return dict(F=F, poses=out, poses_in=poses, log=log, seconds_lm=dt, skipped=skipped)

    def trajectory(self):
        """every scan pushed so far: left the window (final) or still in it (current estimate)"""
        return np.stack(self.done + self.est) if (self.done or self.est) else np.zeros((0, 12))


def main(argv=None):
    import argparse
    from . import capi
    from . import realworld as rw
    ap = argparse.ArgumentParser()
    ap.add_argument("data_dir")
    ap.add_argument("--window", type=int, default=20)
    ap.add_argument("--slide", type=int, default=5)
    ap.add_argument("--voxel", type=float, default=2.0)
    ap.add_argument("--max-scans", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--n-devices", type=int, default=0, help="shard the features over this many GPUs (balm_create_multi)")
    ap.add_argument("--out", default=None, help="write the refined trajectory (N x 12) here as .npy")
    a = ap.parse_args(argv)
    poses, frames = rw.load_window(a.data_dir, a.max_scans or None)
    ctx = capi.Context(a.window, a.device, n_devices=a.n_devices)
    ba = SlidingWindowBA(ctx, a.slide, a.voxel)
    t0 = time.perf_counter()
    for i, (f, p) in enumerate(zip(frames, poses)):
        r = ba.push(f, p)
        if r is not None:
            lg = r["log"]
            print("window %3d (scans %d..%d): %5d features, %2d LM iterations, residual %.6g -> %.6g, %.1f ms%s"
                  % (ba.windows, i - a.window + 1, i, r["F"], len(lg), lg[0, 0] if len(lg) else 0, lg[-1, 1] if len(lg) else 0,
                     1e3 * r["seconds_lm"], "  [not optimised: %s]" % r["skipped"] if r["skipped"] else ""))
    print("%d scans, %d windows in %.2f s" % (len(frames), ba.windows, time.perf_counter() - t0))
    if a.out:
        np.save(a.out, ba.trajectory())
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
