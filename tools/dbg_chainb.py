import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from balm_amd import capi
for W in (208, 216, 300):
    n = 6 * W
    rng = np.random.default_rng(W)
    B = rng.standard_normal((n, 64))
    H = B @ B.T / 64 + np.diag(rng.uniform(0.5, 50.0, n))
    g = rng.standard_normal(n)
    c = capi.Context(W)
    ref = np.linalg.solve(H + 0.1 * np.diag(np.diag(H)), -g)
    os.environ["BALM_SOLVE"] = "chainb"
    os.environ["BALM_CHAINB_IDENT"] = "1"
    dxi, _ = c.solve_damped(H, g, 0.1)
    os.environ.pop("BALM_CHAINB_IDENT")
    print("   with identity rows kept: chainb err %.1e" % (np.abs(dxi - ref).max() / np.abs(ref).max()))
    dx, _ = c.solve_damped(H, g, 0.1)
    os.environ["BALM_SOLVE"] = "chain"
    dx2, _ = c.solve_damped(H, g, 0.1)
    e = np.abs(dx - ref) / np.abs(ref).max()
    # order of elimination = decreasing damped diagonal
    order = np.argsort(-np.abs(np.diag(H)) * 1.1, kind="stable")
    eo = e[order]
    P = (n + 47) // 48
    blocks = [eo[48 * b:48 * b + 48].max() for b in range(P)]
    print("W=%d P=%d chain err %.1e chainb err %.1e; per panel (elimination order) max err:" % (W, P, np.abs(dx2 - ref).max() / np.abs(ref).max(), e.max()),
          " ".join("%.0e" % v for v in blocks))
    c.close()
