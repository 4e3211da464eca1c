#!/usr/bin/env python3
"""Times balm_solve_damped's device part (HIP events, BALM_FLAG_TIMING) for the three factorisation paths at several window
sizes: n = 6 W unknowns.  Usage: python tools/bench_solve.py [W ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balm_amd import capi  # noqa: E402

Ws = [int(a) for a in sys.argv[1:]] or [20, 32, 40, 64, 100, 128, 144, 177, 200, 256, 300, 320, 350, 400, 480, 500, 700, 1024]
for W in Ws:
    n = 6 * W
    rng = np.random.default_rng(W)
    B = rng.standard_normal((n, 64))
    H = B @ B.T / 64 + np.diag(rng.uniform(0.5, 50.0, n))
    g = rng.standard_normal(n)
    c = capi.Context(W, 0, capi.FLAG_TIMING)
    row = []
    for mode in ("launches", "lookahead", "fused", "chain", "chainb"):
        os.environ["BALM_SOLVE"] = mode if mode in ("fused", "chain", "chainb") else "launches"
        os.environ["BALM_LOOKAHEAD"] = "1" if mode == "lookahead" else "0"
        for _ in range(3):
            dx, _ = c.solve_damped(H, g, 0.1)
        c.reset_timing()
        for _ in range(10):
            dx, _ = c.solve_damped(H, g, 0.1)
        ms, cnt = c.timing()["solve"]
        row.append(ms / cnt)
    ref = np.linalg.solve(H + 0.1 * np.diag(np.diag(H)), -g)
    err_b = np.abs(dx - ref).max() / np.abs(ref).max()      # (the last forced mode: chainb)
    os.environ.pop("BALM_SOLVE", None); os.environ.pop("BALM_LOOKAHEAD", None)
    for _ in range(3):
        dx, _ = c.solve_damped(H, g, 0.1)
    c.reset_timing()
    for _ in range(10):
        dx, _ = c.solve_damped(H, g, 0.1)
    ms, cnt = c.timing()["solve"]
    err = np.abs(dx - ref).max() / np.abs(ref).max()
    print("W=%4d n=%5d  launches %.3f ms   + lookahead %.3f ms   fused (r2) %.3f ms   chain (r3) %.3f ms   chain+backsolve %.3f ms   default %.3f ms   err %.1e (backsolve %.1e)"
          % (W, n, row[0], row[1], row[2], row[3], row[4], ms / cnt, err, err_b), flush=True)
    c.close()

if os.environ.get("BALM_SOLVE_TRACE"):
    W = 200
    os.environ["BALM_SOLVE"] = "fused"
    n = 6 * W
    rng = np.random.default_rng(W)
    B = rng.standard_normal((n, 64))
    H = B @ B.T / 64 + np.diag(rng.uniform(0.5, 50.0, n))
    g = rng.standard_normal(n)
    c = capi.Context(W, 0, capi.FLAG_TIMING)
    for _ in range(3):
        c.solve_damped(H, g, 0.1)
    tr = c.solve_trace().astype(np.float64) * 0.01      # us
    P = tr.shape[1]
    t0 = tr[tr > 0].min()
    rhs = tr[P] - t0                                     # the right-hand side block lives through every panel
    print("RHS block, per column: arrive | +wait | +load | +near | +factor | +publish   (us)")
    for p in range(P):
        r = rhs[p]
        print("p=%2d  %8.2f | %6.2f %6.2f %6.2f %6.2f %6.2f" % (p, r[0], r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4]))
    print("owner of block p+1 at column p (its L is what everyone waits for): factor end, publish end")
    for p in range(P - 1):
        r = tr[p + 1][p] - t0
        print("p=%2d  wait %6.2f load %6.2f near %6.2f factor %6.2f publish %6.2f  (ends %8.2f)" % (p, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[5]))
    print("step phases (ns per step, RHS block): publish+barrier | pivot+rows | barrier | operands+MFMA || of pivot+rows: LDS arrival | chain")
    print(np.round(c.step_phases[1:6] * 10.0 / 12.0, 1))
    c.close()
