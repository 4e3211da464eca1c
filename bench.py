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


def cpu_baseline(sc, ctx, target_seconds=18.0, threads=4):
    """CPU baseline leg (rank 0, N=1): the oracle (a port of the reference's Eigen path, compiled
    -std=c++14 -O3 like CMakeLists.txt:8-9) timed on a bounded feature sample of the same
    workload, with the reference's own threading (4 std::threads, bavoxel.hpp:1027)."""
    from oracle import orc
    W, F = sc.W, sc.F
    te, tr = orc.time_sample(0, sc.clusters, None, sc.coeffs, sc.poses_init, min(64, F), threads)
    per_feat = max(te + tr, 1e-6) / min(64, F)
    fs = int(max(64, min(F, target_seconds / per_feat)))
    te, tr = orc.time_sample(0, sc.clusters, None, sc.coeffs, sc.poses_init, fs, threads)
    H, g, _ = ctx.evaluate(0, sc.poses_init)
    ts = orc.time_solve(H, g, 0.1)
    t_iter = (te + tr) * (F / fs) + ts
    extra = {}
    try:    # the reference's own source (oracle/_ref: bavoxel.hpp against the stand-in Eigen), same sample
        from oracle import ref
        if ref.available():
            fr = max(64, fs // 8)
            re_, rr_ = ref.time_sample(sc.clusters, sc.coeffs, sc.poses_init, fr)
            rs_ = ref.time_solve(H, g, 0.1)
            extra["reference_source_iter_per_s"] = 1.0 / ((re_ + rr_) * (F / fr) + rs_)
            extra["reference_source_note"] = ("BALM2::divide_thread_left + evaluate_only_residual of the reference's "
                                              "bavoxel.hpp compiled against oracle/compat (stand-in Eigen, slower than "
                                              "real Eigen), first %d features, 4 threads" % fr)
    except Exception as e:
        extra["reference_source_error"] = repr(e)
    return {
        **extra,
        "value": 1.0 / t_iter, "unit": "iter/s", "cores": threads, "kind": "port",
        "sample": "oracle left_evaluate_acc2 + evaluate_only_residual on the first %d of %d features "
                  "(W=%d), scaled by F/F_sample, + one full %dx%d LDLT solve; %d std::threads as "
                  "bavoxel.hpp:1027" % (fs, F, W, 6 * W, 6 * W, threads),
        "seconds_eval_sample": te, "seconds_resid_sample": tr, "seconds_solve": ts,
        "host_cores": os.cpu_count(),
    }


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
    ap.add_argument("--cpu-seconds", type=float, default=18.0)
    args = ap.parse_args()

    import torch
    from balm_amd import capi, dist as bdist, scene

    rank, local_rank, world = bdist.env_rank()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world),
                  file=sys.stderr)
            sys.exit(2)
    multi = world > 1
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
    if rank != 0:
        return

    F_total = Fg * world
    iters_per_s = args.steps / dt
    value = iters_per_s * (F_total / F_UNIT)      # whole-job aggregate in 50k-feature problem units
    syrk_ms, syrk_n = timing["syrk"]
    syrk_avg_s = syrk_ms / max(syrk_n, 1) * 1e-3
    achieved = wm["syrk_flops_algorithmic"] / syrk_avg_s / 1e12 if syrk_n else None
    roofline = {
        "kernel": "k_hessian_syrk", "bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": (achieved / FP64_PEAK_TFLOPS) if achieved else None, "traffic": None,
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
