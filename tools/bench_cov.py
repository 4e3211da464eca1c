#!/usr/bin/env python3
"""N4 measurement: balm_pose_covariance at the consistency experiment's size (W=100) and at the bench window
(BALM_SYRK=int8 in the environment: the stage's two SYRKs and the Hessian's on the INT8 matrix cores)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from balm_amd import capi, scene

SIZES = ((100, 2000), (200, 20000), (200, 50000))
if len(sys.argv) == 3:                      # one size: python tools/bench_cov.py 200 50000  (kernel tables, PMC passes)
    SIZES = ((int(sys.argv[1]), int(sys.argv[2])),)
for W, F in SIZES:
    sc = scene.generate(9, W, F, 6, mode=1)
    fix = 0.3 * sc.clusters[:, 0]
    fix[:, 9] = np.round(fix[:, 9])
    c = capi.Context(W, flags=capi.FLAG_TIMING)
    c.set_features(sc.clusters, fix, np.ones(F))
    c.pose_covariance(sc.poses_init, point_sigma=0.02, want_raw=False)
    c.reset_timing()
    for _ in range(3):
        c.pose_covariance(sc.poses_init, point_sigma=0.02, want_raw=False)
    t = c.timing()
    print("W=%d F=%d: covariance stage %.2f ms, Hessian %.2f ms" % (W, F, t["cov"][0] / 3, (t["moments"][0] + t["factors"][0] + t["syrk"][0] + t["assemble"][0]) / 3))
    c.close()
