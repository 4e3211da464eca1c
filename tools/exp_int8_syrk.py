#!/usr/bin/env python3
"""BALM_SYRK=int8 against the default FP64 SYRK on the GPU: the Hessian of the full-size evaluation (BASELINE configs[2]: W = 200,
F = 50 000) both ways, against each other and against the reference's own values (tests/golden/lm_big_w200_f50000.npz), and the SYRK span's
time from the library's event timing.  Usage: python tools/exp_int8_syrk.py [--small]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from balm_amd import capi, scene  # noqa: E402


def run(sc, mode, reps):
    if mode:
        os.environ["BALM_SYRK"] = mode
    else:
        os.environ.pop("BALM_SYRK", None)
    os.environ["BALM_SYRK_INT8_MIN_COLS"] = "0"      # (the INT8 product at every size: by itself the switch engages from 12 288 columns on)
    c = capi.Context(sc.W, 0, capi.FLAG_TIMING)
    c.set_features(sc.clusters, None, sc.coeffs)
    H, g, r = c.evaluate(0, sc.poses_init)
    c.reset_timing()
    p = sc.poses_init.copy()
    for k in range(reps):
        p[1, 9] += 1e-9            # (a changed pose: the factor matrix is rebuilt, as in an LM iteration)
        c.evaluate(0, p, want_hess=False)
    ms, cnt = c.timing()["syrk"]
    c.close()
    os.environ.pop("BALM_SYRK", None)
    return H, g, r, ms / max(1, cnt)


def main():
    small = "--small" in sys.argv
    if small:
        for seed, W, F in ((5, 20, 60), (6, 33, 500), (7, 100, 3000), (8, 213, 1000)):
            sc = scene.generate(seed, W, F, 6, mode=1)
            scene.sparsify(sc, seed + 100, 0.2)
            Hd, gd, rd, td = run(sc, "dense", 3)
            Hi, gi, ri, ti = run(sc, "int8", 3)
            scale = np.abs(np.diag(Hd)).max()
            print("W=%d F=%d: max|H_int8 - H_fp64| / max|diag H| = %.3e   g equal %s   syrk %.3f ms vs %.3f ms" % (
                W, F, np.abs(Hi - Hd).max() / scale, np.array_equal(gd, gi), ti, td), flush=True)
        return
    from make_golden_eval import probe_vectors
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "lm_big_w200_f50000.npz")))
    sc = scene.generate(int(g["seed"]), int(g["W"]), int(g["F"]), int(g["pts"]), mode=1)
    t0 = time.time()
    Hd, gd, rd, td = run(sc, None, 20)
    Hi, gi, ri, ti = run(sc, "int8", 20)
    scale = np.abs(g["eval_diag"]).max()
    V = probe_vectors(Hd.shape[0])
    for name, H in (("fp64", Hd), ("int8", Hi)):
        print("%s: diag %.3e  rows %.3e  HV %.3e   (of the reference's values, relative to the largest entry; tolerance 1e-10)" % (
            name, np.abs(np.diag(H) - g["eval_diag"]).max() / scale, np.abs(H[::97] - g["eval_rows"]).max() / scale,
            np.abs(H @ V - g["eval_HV"]).max() / np.abs(g["eval_HV"]).max()))
    print("int8 vs fp64: max|dH| / max|diag H| = %.3e   symmetric %s" % (np.abs(Hi - Hd).max() / scale, np.array_equal(Hi, Hi.T)))
    print("SYRK span (events, per evaluation): fp64 %.3f ms   int8 %.3f ms   [%.1f s]" % (td, ti, time.time() - t0))


if __name__ == "__main__":
    main()
