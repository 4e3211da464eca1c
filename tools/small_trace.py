#!/usr/bin/env python3
"""Phase timeline of k_solve_small (BALM_SOLVE_TRACE=1 must be set before the context is created): python tools/small_trace.py [W ...]"""
import os, sys
import numpy as np
os.environ["BALM_SOLVE_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balm_amd import capi  # noqa: E402
for W in [int(a) for a in sys.argv[1:]] or [8, 16, 20]:
    n = 6 * W
    rng = np.random.default_rng(W)
    B = rng.standard_normal((n, 64))
    H = B @ B.T / 64 + np.diag(rng.uniform(0.5, 50.0, n))
    g = rng.standard_normal(n)
    c = capi.Context(W, 0, capi.FLAG_TIMING)
    for _ in range(5):
        c.solve_damped(H, g, 0.1)
    c.reset_timing()
    c.solve_damped(H, g, 0.1)
    ms, cnt = c.timing()["solve"]
    t = c.solve_trace().ravel()[:18].astype(np.float64) / 100.0        # us (100 MHz wall clock)
    P = (n + 47) // 48
    names = [(1, "rank"), (2, "build")]
    for p in range(P):
        names += [(3 + 4 * p, "p%d factor" % p), (4 + 4 * p, "p%d product" % p)]
        if p + 1 < P:
            names.append((5 + 4 * p, "p%d update" % p))
    names += [(16, "substitute"), (17, "finish")]
    prev = t[0]
    parts = []
    for idx, nm in names:
        parts.append("%s %.2f" % (nm, t[idx] - prev)); prev = t[idx]
    print("W=%d n=%d: kernel %.2f us (HIP events around the launch: %.1f us) | %s" % (W, n, t[17] - t[0], ms / cnt * 1e3, " | ".join(parts)), flush=True)
    c.close()
