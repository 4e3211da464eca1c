#!/usr/bin/env python3
"""k_ldl_chain against LAPACK and against the other factorisation paths, with device times (HIP events).
   python tools/chain_check.py [W ...]     (BALM_SOLVE_TRACE=1: per-panel phases of the chain workgroup at the last W)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balm_amd import capi  # noqa: E402

Ws = [int(a) for a in sys.argv[1:]] or [8, 9, 16, 17, 24, 33, 48, 64, 100, 128, 144, 177, 200, 256, 300, 320]


def matrix(W, kind):
    n = 6 * W
    rng = np.random.default_rng(W * 3 + (kind == "spd"))
    B = rng.standard_normal((n, 64))
    H = B @ B.T / 64 + np.diag(rng.uniform(0.5, 50.0, n))
    if kind == "indefinite":
        s = np.where(rng.uniform(size=n) < 0.2, -1.0, 1.0)
        H = (H * s[:, None]) * s[None, :]
        H[np.diag_indices(n)] *= s
    return H, rng.standard_normal(n)


def timed(c, H, g, mode, reps=10):
    os.environ["BALM_SOLVE"] = mode
    if mode == "launches":
        os.environ["BALM_LOOKAHEAD"] = "1"
    for _ in range(3):
        dx, q1 = c.solve_damped(H, g, 0.1)
    c.reset_timing()
    for _ in range(reps):
        dx, q1 = c.solve_damped(H, g, 0.1)
    ms, cnt = c.timing()["solve"]
    os.environ.pop("BALM_SOLVE", None); os.environ.pop("BALM_LOOKAHEAD", None)
    return dx, q1, ms / cnt


bad = 0
for W in Ws:
    for kind in ("spd", "indefinite"):
        H, g = matrix(W, kind)
        ref = np.linalg.solve(H + 0.1 * np.diag(np.diag(H)), -g)
        c = capi.Context(W, 0, capi.FLAG_TIMING)
        out = []
        for mode in ("chain", "fused", "launches"):
            dx, q1, ms = timed(c, H, g, mode)
            err = np.abs(dx - ref).max() / np.abs(ref).max() if np.all(np.isfinite(dx)) else float("nan")
            out.append("%s %.3f ms err %.1e" % (mode, ms, err))
            if mode == "chain" and not (err < 1e-9):
                bad += 1
        print("W=%4d n=%5d %-10s  %s" % (W, 6 * W, kind, "   ".join(out)), flush=True)
        c.close()
print("chain failures: %d" % bad, flush=True)

if os.environ.get("BALM_SOLVE_TRACE"):
    W = Ws[-1]
    H, g = matrix(W, "spd")
    c = capi.Context(W, 0, capi.FLAG_TIMING)
    os.environ["BALM_SOLVE"] = os.environ.get("CHAIN_TRACE_MODE", "chain")          # "chainb": without identity rows + k_ldl_backsolve
    for _ in range(3):
        c.solve_damped(H, g, 0.1)
    raw = c.solve_trace().astype(np.float64).reshape(-1) * 0.01                                              # us (100 MHz ticks)
    P = (6 * W + 47) // 48
    tr = raw[: P * 16].reshape(-1, 16)
    rw = raw[P * 16: P * 16 + (2 * P + 1) * P * 4].reshape(2 * P + 1, P, 4)
    t0 = tr[0, 0]
    print("chain workgroup, per panel (us): start | chain wave done | riders done | wave 3: far flags seen (x2), L[p+1,p-1] seen, staged, ready | all at B1 | wave 3: product done | B2 | B3 || panel period")
    for p in range(tr.shape[0]):
        r = tr[p]
        nxt = tr[p + 1, 0] - r[0] if p + 1 < tr.shape[0] else float("nan")
        print("p=%2d  %8.2f | %6.2f | %6.2f | %5.2f %5.2f %5.2f %5.2f %5.2f | %6.2f | %6.2f | %6.2f | %6.2f || %6.2f" % (p, r[0] - t0, r[4] - r[0], r[5] - r[0], r[8] - r[0], r[9] - r[0], r[10] - r[0], r[11] - r[0], r[6] - r[0], r[1] - r[0], r[7] - r[1], r[2] - r[1], r[3] - r[2], nxt))

    print("row workgroup rb = p + 2 at column p (its L[p+2, p] feeds the chain's sub-diagonal near update of panel p + 1), us relative to the chain's start of panel p:")
    print("   column begun | inputs ready (tile + near update can go) | Minv_p seen | published   [chain: B1 / B3 of panel p at]")
    for p in range(P - 2):
        r = rw[p + 2, p] - tr[p, 0]
        print("p=%2d  %7.2f | %7.2f | %7.2f | %7.2f    [%.2f / %.2f]" % (p, r[0], r[1], r[2], r[3], tr[p, 1] - tr[p, 0], tr[p, 3] - tr[p, 0]))
