#!/usr/bin/env python3
"""bench.py -- BA iterations/s of the MI355X hot path on synthetic plane clouds.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One *step* = one LM iteration of BALM2::damping_iter (bavoxel.hpp:1104-1157) on the device:
Hessian/gradient evaluation (moments -> factors -> f64-MFMA SYRK -> assemble) + damped LDL^T solve
+ pose update + residual-only evaluation + gain-ratio update.  The Hessian is re-evaluated on every
step (as after an accepted step), never cached.  Inputs are resident in HBM before the timed
region.  Workload at N=1 = BASELINE.json configs[2]: W=200 poses, 50k plane features; for N>1 every
rank holds 50k more features (weak scaling) and the payload is all-reduced over RCCL.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 matrix = vector peak (MI355X_MICROARCH.md: 157.3 TF fp32 / 2)
F_UNIT = 50000              # features per "problem unit" (BASELINE configs[2])


def cpu_baseline(sc, ctx, target_seconds=20.0, threads=4):
    """CPU baseline leg (rank 0, N=1), timed on the box's host cores on a bounded feature sample of
    the same workload and scaled by F/F_sample, plus one full (6W)^2 LDLT solve.  Two candidates,
    both with the reference's own threading (4 std::threads, bavoxel.hpp:1027) and build flags
    (-std=c++14 -O3, CMakeLists.txt:8-9); the faster one is reported as `value`:
      "reference": oracle/_ref -- the reference's bavoxel.hpp compiled against the stand-in Eigen
      "port"     : oracle/balm_oracle.hpp -- the dependency-free restatement"""
    from oracle import orc
    W, F = sc.W, sc.F
    H, g, _ = ctx.evaluate(0, sc.poses_init)
    cands = {}

    def sample_size(per_feat, budget):
        return int(max(64, min(F, budget / max(per_feat, 1e-9))))

    te, tr = orc.time_sample(0, sc.clusters, None, sc.coeffs, sc.poses_init, min(64, F), threads)
    fs = sample_size((te + tr) / min(64, F), 0.5 * target_seconds)
    te, tr = orc.time_sample(0, sc.clusters, None, sc.coeffs, sc.poses_init, fs, threads)
    ts = orc.time_solve(H, g, 0.1)
    cands["port"] = dict(value=1.0 / ((te + tr) * (F / fs) + ts), f_sample=fs, seconds_eval_sample=te,
                         seconds_resid_sample=tr, seconds_solve=ts,
                         what="oracle left_evaluate_acc2 + evaluate_only_residual")
    try:
        from oracle import ref
        if ref.available():
            te, tr = ref.time_sample(sc.clusters, sc.coeffs, sc.poses_init, min(64, F))
            fr = min(sample_size((te + tr) / min(64, F), 0.5 * target_seconds), 6000)   # its evaluator leaks (bavoxel.hpp:312-320)
            te, tr = ref.time_sample(sc.clusters, sc.coeffs, sc.poses_init, fr)
            ts = ref.time_solve(H, g, 0.1)
            cands["reference"] = dict(value=1.0 / ((te + tr) * (F / fr) + ts), f_sample=fr, seconds_eval_sample=te,
                                      seconds_resid_sample=tr, seconds_solve=ts,
                                      what="the reference's BALM2::divide_thread_left + evaluate_only_residual "
                                           "(bavoxel.hpp compiled against oracle/compat's stand-in Eigen)")
    except Exception as e:
        cands["reference_error"] = repr(e)
    kind = max((k for k in ("port", "reference") if k in cands), key=lambda k: cands[k]["value"])
    best = cands[kind]
    out = {
        "value": best["value"], "unit": "iter/s", "cores": threads, "kind": kind,
        "sample": "%s on the first %d of %d features (W=%d), scaled by F/F_sample, + one full %dx%d LDLT solve; "
                  "%d std::threads as bavoxel.hpp:1027" % (best["what"], best["f_sample"], F, W, 6 * W, 6 * W, threads),
        "host_cores": os.cpu_count(), "candidates": cands,
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--win", type=int, default=200, help="W poses")
    ap.add_argument("--features", type=int, default=F_UNIT, help="plane features per GPU")
    ap.add_argument("--pts", type=int, default=6, help="points per (feature, pose)")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    import torch
    from balm_amd import capi, dist as bdist, scene

    rank, local_rank, world = bdist.env_rank()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world),
                  file=sys.stderr)
            sys.exit(2)
    multi = world > 1 or os.environ.get("BALM_BENCH_FORCE_DIST") == "1"   # the latter: exercise the N>1 code path on one GPU
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    if multi:
        bdist.init_process_group("nccl")

    W, Fg = args.win, args.features
    # every rank draws its own 50k-feature shard of one global scene: same trajectory and initial
    # pose noise on all ranks (engine(seed)), disjoint per-feature streams (feature_offset)
    sc = scene.generate(args.seed, W, Fg, args.pts, mode=1, feature_offset=rank * Fg)

    ctx = capi.Context(W, local_rank, capi.FLAG_TIMING)
    ctx.set_features(sc.clusters, None, sc.coeffs)
    if multi:
        bdist.install_allreduce(ctx)

    def barrier():
        torch.cuda.synchronize()
        if multi:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    poses = sc.poses_init
    if args.warmup > 0:
        poses, _ = ctx.damping_iter(poses, form=0, u0=0.1, max_iter=args.warmup, force_hess=True, no_stop=True,
                                    reanchor=False)
    ctx.reset_timing()
    barrier()
    t0 = time.perf_counter()
    poses, lg = ctx.damping_iter(poses, form=0, u0=0.1, max_iter=args.steps, force_hess=True, no_stop=True,
                                 reanchor=False)
    barrier()
    dt = time.perf_counter() - t0
    assert len(lg) == args.steps
    if multi:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    timing = ctx.timing()
    wm = ctx.work_model()
    if multi:
        import torch.distributed as dist
        ctx.set_allreduce(None)
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    F_total = Fg * world
    iters_per_s = args.steps / dt
    value = iters_per_s * (F_total / F_UNIT)      # whole-job aggregate in 50k-feature problem units
    syrk_ms, syrk_n = timing["syrk"]
    syrk_avg_s = syrk_ms / max(syrk_n, 1) * 1e-3
    achieved = wm["syrk_flops_algorithmic"] / syrk_avg_s / 1e12 if syrk_n else None
    traffic = None
    try:   # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
        pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if pj.get("W") == W and pj.get("features_per_gpu") == Fg:
            traffic = pj["k_hessian_syrk"]["hbm_bytes_per_launch"]
    except Exception:
        pass
    roofline = {
        "kernel": "k_hessian_syrk", "bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": (achieved / FP64_PEAK_TFLOPS) if achieved else None, "traffic": traffic,
        "dtype": "f64", "avg_launch_ms": syrk_avg_s * 1e3, "launches": syrk_n,
        "algorithmic_flops_per_launch": wm["syrk_flops_algorithmic"],
        "issued_flops_per_launch": wm["syrk_flops_issued"],
    }
    out = {
        "metric": "BA iterations/sec (W poses x F plane features)",
        "value": value, "unit": "iter/s per 50k-feature problem unit",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: W=%d poses, %d plane features per GPU (%d total), "
                               "%d pts per (feature,pose), full HIP accumulate + LDL^T LM solve"
                               % (W, Fg, F_total, args.pts),
                   "W": W, "features_per_gpu": Fg, "features_total": F_total, "form": "left",
                   "parallelism": "features sharded x%d, RCCL all-reduce of [tiles|blockdiag|r]" % world
                   if multi else "single GPU"},
        "lm_iterations_per_sec_raw": iters_per_s,
        "kernel_ms_per_step": {k: v[0] / args.steps for k, v in timing.items()},
        "roofline": roofline,
        "final_residual": float(lg[-1, 1]),
    }
    if world == 1 and not args.no_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline(sc, ctx, args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:   # the baseline leg must never take the GPU number down with it
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
