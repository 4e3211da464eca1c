#!/usr/bin/env python3
"""Per-kernel timing of one LM iteration on the shipped real-world window's shape (W=177, F=2281, 15 % block fill),
dense vs block-sparse hessian_syrk plan, both factorisation paths.  Uses the fixture oracle/_ref/realworld_features.npz
(the reference's own association on datas/benchmark_realworld) when present, a banded synthetic stand-in otherwise."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balm_amd import capi, scene  # noqa: E402

fx = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "realworld_features.npz")
if os.path.exists(fx):
    g = np.load(fx)
    cl, co, poses = g["clusters"], g["coeffs"], g["poses"]
    what = "shipped window (reference association)"
else:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_gpu_sparse_syrk import banded_scene
    sc = banded_scene(1, 177, 2281, 5, 4, 40)
    cl, co, poses = sc.clusters, sc.coeffs, sc.poses_init
    what = "banded synthetic stand-in"
F, W = cl.shape[:2]
na = (cl[..., 9] > 0).sum(1)
print("%s: W=%d F=%d S=%d fill=%.1f%%" % (what, W, F, na.sum(), 100.0 * (na * (na + 1) / 2).sum() / (F * W * (W + 1) / 2)))
for syrk in ("dense", "sparse"):
    for solve in ("launches", "fused", "default"):       # default at this window: k_ldl_chain (round 3)
        os.environ["BALM_SYRK"] = syrk
        os.environ["BALM_SOLVE"] = solve
        if solve == "default":
            os.environ.pop("BALM_SOLVE")
        c = capi.Context(W, 0, capi.FLAG_TIMING)
        c.set_features(cl, None, co)
        c.damping_iter(poses, u0=0.01, max_iter=3, force_hess=True, no_stop=True, reanchor=False)
        c.reset_timing()
        import time
        t0 = time.perf_counter()
        K = 20
        c.damping_iter(poses, u0=0.01, max_iter=K, force_hess=True, no_stop=True, reanchor=False)
        dt = (time.perf_counter() - t0) / K * 1e3
        t = c.timing()
        wm = c.work_model()
        print("syrk %-6s solve %-8s: %.3f ms/step | " % (syrk, solve, dt) + "  ".join("%s %.3f" % (k, v[0] / K) for k, v in t.items() if v[1])
              + " | issued/algorithmic flops %.2f" % (wm["syrk_flops_issued"] / wm["syrk_flops_algorithmic"]), flush=True)
        c.close()
