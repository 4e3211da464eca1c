"""BALM_SOLVE_TRACE=1: what the helper workgroups of k_ldl_chain did during one factorisation (updates, busy and idle ticks)"""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["BALM_SOLVE_TRACE"] = "1"
import numpy as np
from balm_amd import capi
for W, mode in ((200, "chain"), (300, "chainb"), (400, "chainb"), (500, "chainb")):
    n = 6 * W
    rng = np.random.default_rng(W)
    B = rng.standard_normal((n, 64))
    H = B @ B.T / 64 + np.diag(rng.uniform(0.5, 50.0, n))
    g = rng.standard_normal(n)
    c = capi.Context(W, 0, capi.FLAG_TIMING)
    os.environ["BALM_SOLVE"] = mode
    for _ in range(3):
        c.solve_damped(H, g, 0.1)
    c.reset_timing()
    c.solve_damped(H, g, 0.1)
    ms = c.timing()["solve"][0]
    P = (n + 47) // 48
    RB = 2 * P + 1
    cap = RB * P * 6 + 6 * P
    buf = (C.c_longlong * cap)()
    dims = (C.c_int * 3)()
    c._check(c.L.balm_get_solve_trace(c.h, buf, cap, dims))
    t = np.frombuffer(buf, dtype=np.int64)
    off = P * 16 + RB * P * 4
    NH = 256 - 1 - (RB if mode == "chain" else P + 1)
    macro = mode == "chainb" and P >= 31 and (P <= 41 or P >= 69)       # launch_factor_chain's rule: 2 x 2 macro-tiles at 31..41 and 69..100 panels
    M = sum((P - j0 + 2) // 2 for j0 in range(2, P, 2))
    NH = min(NH, (P - 2) * P if mode == "chain" else (M if macro else P * (P - 1) // 2 - 1))
    st = t[off:off + 4 * NH].reshape(NH, 4)
    upd, busy, idle = st[:, 0], st[:, 1] / 100.0, st[:, 2] / 100.0      # 100 MHz ticks -> us
    load, mfma = (st[:, 3] >> 32) / 100.0, (st[:, 3] & 0xffffffff) / 100.0
    ch = t[:P * 16].reshape(P, 16)
    print("W=%d n=%d P=%d %s: solve %.3f ms; %d helpers: updates %d total (%.0f..%.0f per helper), busy %.0f us avg (%.2f us per update), idle %.0f us avg, max busy %.0f us"
          % (W, n, P, mode, ms, NH, upd.sum(), upd.min(), upd.max(), busy.mean(), busy.sum() / max(upd.sum(), 1), idle.mean(), busy.max()))
    print("      per update: %.2f us until tile + operands are in LDS, %.2f us MFMAs + choice of the next job, %.2f us store / publish"
          % (load.sum() / max(upd.sum(), 1), mfma.sum() / max(upd.sum(), 1), (busy.sum() - load.sum() - mfma.sum()) / max(upd.sum(), 1)))
    c.close()
