#!/usr/bin/env python3
"""pmc_summary.csv (tools/pmc_summary.py over the rocprofv3 --pmc passes of `bench.py --steps 3 --warmup 1 --no-cpu`)
-> profiles/pmc_traffic.json, the per-launch HBM traffic bench.py reports as roofline.traffic.
Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes for gfx950: FETCH_SIZE is in KiB and
tallies 128-byte requests at 64 bytes (x 1024 x 2); WRITE_SIZE is in KiB (x 1024).
Usage: make_pmc_traffic.py <pmc_summary.csv> <tag> [W features_per_gpu]"""
import csv
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
W = int(sys.argv[3]) if len(sys.argv) > 3 else 200
Fg = int(sys.argv[4]) if len(sys.argv) > 4 else 50000
rows = {r["kernel"].split("::")[-1].split("<")[0]: r for r in csv.DictReader(open(src))}
out = {"run": tag, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE "
                 "(separate passes) -- python bench.py --steps 3 --warmup 1 --no-cpu; profiles/%s_pmc_summary.csv" % tag,
       "correction": "FETCH_SIZE (KiB) x 1024 x 2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE (KiB) x 1024",
       "W": W, "features_per_gpu": Fg}
# k_hessian_syrk: what the ALGORITHM has to move is G-tilde read once (3F columns x 6W rows) + one Hessian written (the upper tiles); the
# split-K partial tiles (35 slices x 114 jobs x 6 400 doubles, written here and re-read by the reduce) are this implementation's own
# traffic and are listed beside it, not inside it
alg = {"k_hessian_syrk": 8.0 * (3.0 * Fg) * (6.0 * W) + 8.0 * 6400 * 114,
       "k_feature_factors": 224.0 * Fg * W, "k_world_moments": 80.0 * Fg * W}
extra = {"k_hessian_syrk": {"split_k_partial_tile_bytes": 8.0 * 6400 * 35 * 114}}
for k in ("k_hessian_syrk", "k_feature_factors", "k_world_moments", "k_ldl_chain", "k_ldl_fused", "k_build_clusters_runs"):
    r = rows.get(k)
    if not r or not r.get("FETCH_SIZE"):
        continue
    f = float(r["FETCH_SIZE"]) * 1024 * 2
    w = float(r["WRITE_SIZE"] or 0) * 1024
    e = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes_per_launch": f + w}
    if r.get("SQ_BUSY_CU_CYCLES") and float(r["SQ_BUSY_CU_CYCLES"]) > 0:
        e["mfma_busy_frac"] = float(r.get("SQ_VALU_MFMA_BUSY_CYCLES") or 0) / float(r["SQ_BUSY_CU_CYCLES"]) / 4.0
    if r.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE comes out summed over the 8 XCDs of the device: per-XCD cycles / duration = the clock
        e["clock_ghz_under_pmc"] = float(r["GRBM_GUI_ACTIVE"]) / 8.0 / float(r["dur_ns"])
    if k in alg:
        e["algorithmic_bytes"] = alg[k]
        e["traffic_over_algorithmic"] = (f + w) / alg[k]
    e.update(extra.get(k, {}))
    out[k] = e
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, {k: round(v["hbm_bytes_per_launch"] / 1e9, 3) for k, v in out.items() if isinstance(v, dict)})
