"""Extreme shapes through the C ABI (many features x few poses, the 1024-pose limit, ...): size-independent properties + an LM run."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from balm_amd import capi, scene
for W, F in ((20, 1000000), (1024, 3000), (64, 200000)):
    sc = scene.generate(3, W, F, 4, mode=1)
    c = capi.Context(W, 0, capi.FLAG_TIMING)
    c.set_features(sc.clusters, None, sc.coeffs)
    h = F // 3
    H, g, r = c.evaluate(0, sc.poses_init)
    H1, g1, r1 = c.evaluate(0, sc.poses_init, 0, h)
    H2, g2, r2 = c.evaluate(0, sc.poses_init, h, F)
    e = np.abs(H1 + H2 - H).max() / np.abs(H).max()
    t = time.time()
    out, lg = c.damping_iter(sc.poses_init, form=0, u0=0.1, max_iter=20)
    dt = time.time() - t
    print("W=%d F=%d: subrange additivity %.1e, sym %s, LM %d iters %.1f ms, residual %.4g -> %.4g" % (W, F, e, np.array_equal(H, H.T), len(lg), dt * 1e3, lg[0, 0], lg[-1, 1]), flush=True)
    c.close()
