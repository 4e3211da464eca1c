#!/usr/bin/env python3
"""The reference's OWN Hessian / gradient / residual at the bench size, pinned for a direct comparison (not through sub-ranges
or LM traces): BALM2::divide_thread_left (src/benchmark/bavoxel.hpp:1025-1059 -> VOX_HESS::left_evaluate_acc2 :304-426, four
std::threads) over ALL features of BASELINE configs[2] (W = 200, F = 50 000, bench.py's scene, seed 2024) at its initial poses.

    python tests/golden/make_golden_eval.py        # needs /root/reference (build container only); ~15 s, leaks reclaimed

The 11.5 MB Hessian itself is not stored; what is: H V for eight fixed pseudo-random vectors V (any wrong entry of H moves some
product), diag(H), every 97th row of H, g and the residual -> merged into tests/golden/lm_big_w200_f50000.npz as eval_*."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from balm_amd import scene  # noqa: E402
from oracle import ref  # noqa: E402
from make_golden_big import CASES, checksums  # noqa: E402


def probe_vectors(n, k=8):
    return np.random.default_rng(20250924).standard_normal((n, k))


def main():
    os.environ.setdefault("BALM_REF_RECLAIM_LEAKS", "1")
    ref.build()
    name = "lm_big_w200_f50000"
    c = CASES[name]
    path = os.path.join(HERE, name + ".npz")
    out = dict(np.load(path))
    sc = scene.generate(c["seed"], c["W"], c["F"], c["pts"], mode=1)
    assert np.array_equal(checksums(sc), out["checksums"]), "the scene generator drifted"
    t0 = time.time()
    H, g, r = ref.divide_thread(0, sc.clusters, None, sc.coeffs, sc.poses_init)
    print("reference divide_thread_left at W=%d F=%d: %.1f s, residual %.9g" % (c["W"], c["F"], time.time() - t0, r), flush=True)
    # (off-diagonal blocks are mirrored by :422-424; the 6x6 diagonal blocks are symmetric only to rounding)
    assert np.abs(H - H.T).max() <= 1e-12 * np.abs(H).max()
    V = probe_vectors(H.shape[0])
    out.update(eval_HV=H @ V, eval_diag=np.diag(H).copy(), eval_rows=H[::97].copy(), eval_g=g, eval_r=r)
    np.savez_compressed(path, **out)
    print("->", path, "%.2f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
