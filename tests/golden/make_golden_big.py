#!/usr/bin/env python3
"""Golden LM traces + final poses at the BASELINE sizes themselves -- the north star's acceptance test
(final poses within 1e-5 rad / 1e-4 m of the reference CPU optimizer) pinned at its own size.

    python tests/golden/make_golden_big.py [case ...]     # needs /root/reference (build container only)

Both of the reference's optimizers are run, each from ITS OWN SOURCE compiled where it lies (oracle/ref_build.sh):
  virtual  BALM2::dampingIter(x_stats, plSurfs)  src/benchmark/benchmark_virtual.cpp:375-482   (u0 = 0.1, <= 20 iterations,
           clusters pushed from the float point clouds :392-403, weights winSize*ptsSize :391, single thread,
           pose 0 -> identity at the end :472-479)                         -> oracle/_ref/libbalm_ref_virtual.so
  bavoxel  BALM2::damping_iter(x_stats, voxhess) src/benchmark/bavoxel.hpp:1069-1166            (u0 = 0.01, <= 10 iterations,
           4 std::threads, >= 20 planes per pose, re-anchor :1159-1164)    -> oracle/_ref/libbalm_ref.so
Inputs are NOT stored (800 MB at W=200/F=50k): they regenerate from the seed with balm_amd.scene.generate(mode=1);
the fixture carries checksums of the regenerated inputs so that a drifted generator fails loudly instead of
silently comparing different problems.  Outputs per case: lm_log_* rows (r1 r2 u v q q1 accepted, parsed from the
reference's printf line: 6 decimals), lm_poses_* [W,12], seconds_* (this container's CPU).

Cases = BASELINE.json configs[1] (W=64, F=5 000, ~2 M points) and configs[2] (W=200, F=50 000; the bench workload,
same seed as bench.py).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from balm_amd import scene  # noqa: E402
from oracle import ref, ref_virtual  # noqa: E402

CASES = {
    "lm_big_w64_f5000": dict(seed=2025, W=64, F=5000, pts=6),
    "lm_big_w200_f50000": dict(seed=2024, W=200, F=50000, pts=6),      # bench.py's scene
    # BASELINE configs[3]: the 8-GPU problem -- the SAME global scene bench.py --gpus N shards (seed 2024, mode 1: feature a
    # has its own engine, so rank r's shard is features [r F/N, (r+1) F/N) of this table) and its strong_scaling_reference.  bavoxel flavour only by
    # default (4 threads, ~6 min here; needs BALM_REF_RECLAIM_LEAKS=1, set below: bavoxel.hpp:312-320 leaks 20 GB per
    # evaluation at this size); `virtual` (single thread, ~25 min, 240 M points) when asked for: "lm_big_w200_f200000:virtual"
    "lm_big_w200_f200000": dict(seed=2024, W=200, F=200000, pts=6, flavours=("bavoxel",)),
    # BASELINE configs[4] names a 500-pose window: the factorisation path of windows above 320 poses (launches with
    # lookahead) under an LM run with sparse co-visibility.  `bavoxel` = the reference's optimizer (u0 = 0.01, 10 it.);
    # `oracle_u01` = the oracle's LM loop with benchmark_virtual.cpp's constants (u0 = 0.1, 20 it.; the reference's own
    # dampingIter(x_stats, plSurfs) builds dense clusters from clouds and cannot take a sparse table)
    "lm_big_sparse_w500_f2000": dict(seed=31, W=500, F=2000, pts=6, drop=0.5, flavours=("bavoxel", "oracle_u01")),
}


def checksums(sc):
    """position-weighted sums: sensitive to any permuted or perturbed entry, cheap to recompute"""
    cl = sc.clusters.reshape(-1)
    w = (np.arange(cl.size, dtype=np.float64) % 977.0) + 1.0
    return np.array([cl.sum(), float(np.dot(cl, w)), sc.poses_init.sum(), sc.coeffs.sum(),
                     float(np.dot(sc.poses_init.reshape(-1), np.arange(1, sc.poses_init.size + 1)))])


def main():
    os.environ.setdefault("BALM_REF_RECLAIM_LEAKS", "1")        # read once by oracle/compat/Eigen/Core when _ref loads
    ref.build()
    names = sys.argv[1:] or [n for n in CASES if "flavours" not in CASES[n]]
    for name in names:
        name, _, extra = name.partition(":")
        c = CASES[name]
        flavours = tuple(c.get("flavours", ("virtual", "bavoxel"))) + ((extra,) if extra else ())
        path = os.path.join(HERE, name + ".npz")
        sc = scene.generate(c["seed"], c["W"], c["F"], c["pts"], mode=1, keep_points="virtual" in flavours)
        if c.get("drop"):
            scene.sparsify(sc, c["seed"] + 100, c["drop"])          # as tests/util.make_scene
        out = dict(np.load(path)) if os.path.exists(path) and extra else {}
        out.update(seed=c["seed"], W=c["W"], F=c["F"], pts=c["pts"], drop=c.get("drop", 0.0), checksums=checksums(sc), poses_gt=sc.poses_gt)
        if "virtual" in flavours:
            t0 = time.time()
            poses, lg, sec = ref_virtual.damping_iter(sc.points, sc.poses_init)
            out["lm_poses_virtual"], out["lm_log_virtual"], out["seconds_virtual"] = poses, lg, sec
            print(name, "virtual: %d iterations, %.1f s (%.1f s wall)" % (len(lg), sec, time.time() - t0), flush=True)
            print(lg[:, :3], flush=True)
        if "bavoxel" in flavours and not (extra and "lm_poses_bavoxel" in out):
            t0 = time.time()
            poses, lg = ref.damping_iter(sc.clusters, None, sc.coeffs, sc.poses_init)
            out["lm_poses_bavoxel"], out["lm_log_bavoxel"], out["seconds_bavoxel"] = poses, lg, time.time() - t0
            print(name, "bavoxel: %d iterations, %.1f s" % (len(lg), time.time() - t0), flush=True)
            print(lg[:, :3], flush=True)
        if "oracle_u01" in flavours:
            from oracle import orc
            t0 = time.time()
            poses, lg = orc.damping_iter(0, sc.clusters, None, sc.coeffs, sc.poses_init, 0.1, 20, threads=8)
            out["lm_poses_oracle_u01"], out["lm_log_oracle_u01"], out["seconds_oracle_u01"] = poses, lg, time.time() - t0
            print(name, "oracle (u0 = 0.1, <= 20 it.): %d iterations, %.1f s" % (len(lg), time.time() - t0), flush=True)
            print(lg[:, :3], flush=True)
        np.savez_compressed(path, **out)


if __name__ == "__main__":
    main()
