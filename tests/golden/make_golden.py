#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by running the REFERENCE'S OWN SOURCE
(/root/reference/include/tools.hpp + src/benchmark/bavoxel.hpp, compiled by oracle/ref_build.sh
against the stand-in headers of oracle/compat/) on seeded synthetic scenes.

    python tests/golden/make_golden.py        # needs /root/reference (build container only)

Each .npz holds the inputs (clusters [F,W,10], coeffs [F], poses [W,12]) and the reference's outputs:
  H0,g0,r0  VOX_HESS::left_evaluate_acc2      H1,g1,r1  VOX_HESS::acc_evaluate2
  H2        VOX_HESS::left_evaluate (the un-accelerated left form)      (H1, H2 only for W <= 20)
  r_only    VOX_HESS::evaluate_only_residual
  dx_u*,q1_u*  (Hess + u*D).ldlt().solve(-JacT) and q1 at u = 0.01, 0.1 on (H0, g0)
  lm_poses, lm_log  BALM2::damping_iter (left form, u0 = 0.01, <= 10 iterations); log rows parsed from
                    the reference's printf line (6 decimals)
Fix clusters are empty, as in benchmark_realworld (the only driver of bavoxel.hpp).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref  # noqa: E402
from util import make_scene  # noqa: E402

CASES = {
    "virtual_w20_f20": dict(seed=1, W=20, F=20, pts=40, drop=0.0),     # launch/benchmark_virtual.launch sizes
    "sparse_w7_f9": dict(seed=2, W=7, F=9, pts=12, drop=0.3),
    "sparse_w33_f60": dict(seed=3, W=33, F=60, pts=8, drop=0.5),
}


def main():
    ref.build()
    for name, c in CASES.items():
        sc, _ = make_scene(c["seed"], c["W"], c["F"], c["pts"], c["drop"])
        out = dict(clusters=sc.clusters, coeffs=sc.coeffs, poses=sc.poses_init, poses_gt=sc.poses_gt)
        for form in (0, 1):
            H, g, r = ref.evaluate(form, sc.clusters, None, sc.coeffs, sc.poses_init)
            out["g%d" % form], out["r%d" % form] = g, r
            if form == 0 or c["W"] <= 20:          # keep the fixtures small
                out["H%d" % form] = H
        if c["W"] <= 20:
            out["H2"] = ref.evaluate(2, sc.clusters, None, sc.coeffs, sc.poses_init)[0]
        out["r_only"] = ref.only_residual(sc.clusters, None, sc.coeffs, sc.poses_init)
        for u in (0.01, 0.1):
            dx, q1 = ref.solve_damped(out["H0"], out["g0"], u)
            out["dx_u%g" % u], out["q1_u%g" % u] = dx, q1
        # damping_iter exit(0)s below 20 planes per pose (bavoxel.hpp:1079-1085): only run it where safe
        if (sc.clusters[..., 9] > 0).sum(0).min() >= 20:
            poses, lg = ref.damping_iter(sc.clusters, None, sc.coeffs, sc.poses_init)
            out["lm_poses"], out["lm_log"] = poses, lg
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
