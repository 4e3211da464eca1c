#!/usr/bin/env python3
"""a few solves at one window size (for tools/gpu_solve_trace.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from balm_amd import capi
W = int(sys.argv[1]); n = 6 * W
rng = np.random.default_rng(W)
B = rng.standard_normal((n, 64))
H = B @ B.T / 64 + np.diag(rng.uniform(0.5, 50.0, n))
g = rng.standard_normal(n)
c = capi.Context(W)
for _ in range(4):
    c.solve_damped(H, g, 0.1)
c.close()
