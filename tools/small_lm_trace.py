#!/usr/bin/env python3
"""One LM run of a small window for `rocprofv3 --kernel-trace`: which kernels an iteration launches (tools/gpu_r04w.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balm_amd import capi, scene
W, F = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 20)
sc = scene.generate(1, W, F, 40, mode=1)
c = capi.Context(W)
c.set_features(sc.clusters, None, sc.coeffs)
for _ in range(3):
    c.damping_iter(sc.poses_init, u0=0.1, max_iter=20, force_hess=True, no_stop=True, reanchor=False)
c.close()
