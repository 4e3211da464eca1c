#!/usr/bin/env python3
"""bench.py -- BA iterations/s of the MI355X hot path on synthetic plane clouds.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One *step* = one LM iteration of BALM2::damping_iter (bavoxel.hpp:1104-1157) on the device:
Hessian/gradient evaluation (moments -> factors -> f64-MFMA SYRK -> assemble) + damped LDL^T solve
+ pose update + residual-only evaluation + gain-ratio update.  The Hessian is re-evaluated on every
step (as after an accepted step), never cached.  Inputs are resident in HBM before the timed region.

Workloads (BASELINE.json):  N = 1  -> configs[2]: W=200 poses, 50 000 plane features on the one GPU.
                            N > 1  -> configs[3]: W=200 poses, 200 000 plane features IN TOTAL, sharded over the N
                                      ranks (25 000 per GPU at N = 8), summed by one RCCL all-reduce per evaluation
                                      issued inside libbalm_hip.so on its own stream; `value` is the plain
                                      iterations/s of that one problem ("scaling": "strong" over N = 2, 4, 8).
                                      --weak keeps 50 000 features per GPU instead.
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
HBM_PEAK_TBS = 8.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
F_SINGLE = 50000            # BASELINE configs[2]
F_SHARDED_TOTAL = 200000    # BASELINE configs[3]


def cpu_baseline(sc, ctx, target_seconds=20.0):
    """CPU baseline leg (rank 0, N=1), timed on the box's host cores on a bounded feature sample of the same
    workload and scaled by F/F_sample, plus one full (6W)^2 LDLT solve.  Build flags are the reference's own
    (-std=c++14 -O3, CMakeLists.txt:8-9).  Candidates:
      "reference"          oracle/_ref: the reference's bavoxel.hpp, BALM2::divide_thread_left with its 4 std::threads
                           (bavoxel.hpp:1027) -- the form BASELINE's real-world driver runs
      "reference_virtual"  oracle/_ref: the reference's benchmark_virtual.cpp, BALM2::left_evaluate_acc2 on ONE thread
                           (benchmark_virtual.cpp:413) -- the form BASELINE configs[0..3] name
      "port"               oracle/balm_oracle.hpp, the dependency-free restatement, 4 threads
      "port_all_cores"     the same on every host core
    The reference's sources are compiled against oracle/compat's STAND-IN Eigen (plain loops, no expression
    templates or SIMD kernels: real Eigen would be faster, it is not in this image).  `value` = the fastest
    candidate that uses the reference's own threading (4 threads or fewer); the all-cores figure is listed only."""
    from oracle import orc
    W, F = sc.W, sc.F
    H, g, _ = ctx.evaluate(0, sc.poses_init)
    cands = {}
    share = target_seconds / 4.0

    def sample_size(per_feat, budget):
        return int(max(64, min(F, budget / max(per_feat, 1e-9))))

    def port(threads, key):
        te, tr = orc.time_sample(0, sc.clusters, None, sc.coeffs, sc.poses_init, min(64, F), threads)
        fs = sample_size((te + tr) / min(64, F), 0.5 * share)
        te, tr = orc.time_sample(0, sc.clusters, None, sc.coeffs, sc.poses_init, fs, threads)
        ts = orc.time_solve(H, g, 0.1)
        cands[key] = dict(value=1.0 / ((te + tr) * (F / fs) + ts), f_sample=fs, threads=threads, seconds_eval_sample=te,
                          seconds_resid_sample=tr, seconds_solve=ts, what="oracle left_evaluate_acc2 + evaluate_only_residual")

    port(4, "port")
    port(os.cpu_count() or 1, "port_all_cores")
    try:
        from oracle import ref
        if ref.available():
            te, tr = ref.time_sample(sc.clusters, sc.coeffs, sc.poses_init, min(64, F))
            fr = min(sample_size((te + tr) / min(64, F), 0.5 * share), 6000)   # its evaluator leaks (bavoxel.hpp:312-320)
            te, tr = ref.time_sample(sc.clusters, sc.coeffs, sc.poses_init, fr)
            ts = ref.time_solve(H, g, 0.1)
            cands["reference"] = dict(value=1.0 / ((te + tr) * (F / fr) + ts), f_sample=fr, threads=4, seconds_eval_sample=te,
                                      seconds_resid_sample=tr, seconds_solve=ts,
                                      what="the reference's BALM2::divide_thread_left + evaluate_only_residual "
                                           "(bavoxel.hpp compiled against oracle/compat's stand-in Eigen)")
    except Exception as e:
        cands["reference_error"] = repr(e)
    try:
        from oracle import ref, ref_virtual
        if ref_virtual.available():
            def tv(fs):
                t0 = time.perf_counter()
                ref_virtual.evaluate(0, sc.clusters[:fs], None, sc.coeffs[:fs], sc.poses_init)
                t1 = time.perf_counter()
                ref_virtual.evaluate(3, sc.clusters[:fs], None, sc.coeffs[:fs], sc.poses_init)
                return t1 - t0, time.perf_counter() - t1
            te, tr = tv(min(32, F))
            fv = min(sample_size((te + tr) / min(32, F), 0.5 * share), 4000)
            te, tr = tv(fv)
            ts = ref.time_solve(H, g, 0.1) if ref.available() else orc.time_solve(H, g, 0.1)
            cands["reference_virtual"] = dict(value=1.0 / ((te + tr) * (F / fv) + ts), f_sample=fv, threads=1, seconds_eval_sample=te,
                                              seconds_resid_sample=tr, seconds_solve=ts,
                                              what="the reference's benchmark_virtual.cpp BALM2::left_evaluate_acc2 + only_residual, "
                                                   "single thread as :413 (compiled against oracle/compat's stand-in Eigen)")
    except Exception as e:
        cands["reference_virtual_error"] = repr(e)
    eligible = [k for k in ("port", "reference", "reference_virtual") if k in cands]
    kind_key = max(eligible, key=lambda k: cands[k]["value"])
    best = cands[kind_key]
    return {
        "value": best["value"], "unit": "iter/s", "cores": best["threads"],
        "kind": "reference" if kind_key.startswith("reference") else "port", "candidate": kind_key,
        "sample": "%s on the first %d of %d features (W=%d), scaled by F/F_sample, + one full %dx%d LDLT solve; %d thread(s); "
                  "linear algebra = oracle/compat's stand-in Eigen (real Eigen is not in this image and would be faster)"
                  % (best["what"], best["f_sample"], F, W, 6 * W, 6 * W, best["threads"]),
        "host_cores": os.cpu_count(), "candidates": cands,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--win", type=int, default=200, help="W poses")
    ap.add_argument("--features", type=int, default=0, help="plane features per GPU (default: configs[2] / configs[3])")
    ap.add_argument("--weak", action="store_true", help="N > 1: 50 000 features per GPU instead of 200 000 in total")
    ap.add_argument("--pts", type=int, default=6, help="points per (feature, pose)")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--no-cpu", action="store_true", help="skip the extra legs (cpu_baseline, strong-scaling reference): profiling runs")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-strong-ref", action="store_true",
                    help="N = 1: skip the extra untimed-contract leg that runs configs[3]'s 200 000 features on the one GPU")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "hook"],
                    help="N > 1: RCCL inside the library (stream-ordered) or the torch.distributed hook")
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

    W = args.win
    if args.features > 0:
        Fg = args.features
    elif world > 1 and not args.weak:
        Fg = F_SHARDED_TOTAL // world
    else:
        Fg = F_SINGLE
    # every rank draws its own shard of one global scene: same trajectory and initial pose noise on all ranks
    # (engine(seed)), disjoint per-feature streams (feature_offset)
    sc = scene.generate(args.seed, W, Fg, args.pts, mode=1, feature_offset=rank * Fg)

    ctx = capi.Context(W, local_rank, capi.FLAG_TIMING)
    ctx.set_features(sc.clusters, None, sc.coeffs)
    transport = None
    if multi:
        if args.transport == "rccl":
            transport = bdist.install_rccl(ctx)
        else:
            bdist.install_allreduce(ctx)
            transport = "torch.distributed hook (host-synchronised)"

    def barrier():
        torch.cuda.synchronize()
        if multi:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # the natural LM run first (untimed region of the contract; reported as its own figure): from the noisy start to
    # the reference's own stop rule, benchmark_virtual.cpp's constants
    barrier()
    t0 = time.perf_counter()
    _, lg_nat = ctx.damping_iter(sc.poses_init, form=0, u0=0.1, max_iter=20)
    barrier()
    t_nat = time.perf_counter() - t0

    def run_steps(k):
        """exactly k LM iterations, in runs of at most 20 from the noisy start (a run that sat at the optimum for hundreds
        of iterations would only multiply the damping after rounding-level rejections)"""
        logs = []
        while k > 0:
            m = min(20, k)
            _, lg_ = ctx.damping_iter(sc.poses_init, form=0, u0=0.1, max_iter=m, force_hess=True, no_stop=True, reanchor=False)
            assert len(lg_) == m
            logs.append(lg_)
            k -= m
        return np.concatenate(logs)

    if args.warmup > 0:
        run_steps(args.warmup)
    ctx.reset_timing()
    barrier()
    t0 = time.perf_counter()
    lg = run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    assert len(lg) == args.steps
    if multi:
        import torch.distributed as dist
        tt = torch.tensor([dt, t_nat], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, t_nat = float(tt[0].item()), float(tt[1].item())

    timing = ctx.timing()
    wm = ctx.work_model()
    if multi:
        import torch.distributed as dist
        if args.transport == "hook":
            ctx.set_allreduce(None)
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    F_total = Fg * world
    iters_per_s = args.steps / dt
    n = 6 * W
    per_step = {k: v[0] / args.steps for k, v in timing.items()}

    def avg_s(key):
        ms, cnt = timing[key]
        return ms / max(cnt, 1) * 1e-3 if cnt else None

    syrk_s = avg_s("syrk")
    achieved = wm["syrk_flops_algorithmic"] / syrk_s / 1e12 if syrk_s else None
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
        "dtype": "f64", "avg_launch_ms": syrk_s * 1e3 if syrk_s else None, "launches": timing["syrk"][1],
        "algorithmic_flops_per_launch": wm["syrk_flops_algorithmic"],
        "issued_flops_per_launch": wm["syrk_flops_issued"],
    }
    # the other kernel classes of the step against THEIR rooflines (algorithmic bytes/flops of SURVEY 8d; S = observations)
    S = wm["S"]
    secondary = {}
    t = avg_s("moments")
    if t:      # K1 + K1b: 80 B per observation read (two evaluations per step: Hessian side and residual side)
        secondary["moments"] = {"bound": "hbm", "achieved": 80.0 * S / t / 1e12, "peak": HBM_PEAK_TBS, "unit": "TB/s",
                                "frac": 80.0 * S / t / 1e12 / HBM_PEAK_TBS, "avg_launch_ms": t * 1e3}
    t = avg_s("factors")
    if t:      # K2: 80 B read + 144 B written per observation
        secondary["factors"] = {"bound": "hbm", "achieved": 224.0 * S / t / 1e12, "peak": HBM_PEAK_TBS, "unit": "TB/s",
                                "frac": 224.0 * S / t / 1e12 / HBM_PEAK_TBS, "avg_launch_ms": t * 1e3}
    t = avg_s("solve")
    if t:      # blocked LDL^T: n^3/3 + 2 n^2 flops; a latency chain, priced against the FP64 peak for the record
        fl = n ** 3 / 3.0 + 2.0 * n * n
        secondary["solve"] = {"bound": "latency", "achieved": fl / t / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": fl / t / 1e12 / FP64_PEAK_TFLOPS, "avg_launch_ms": t * 1e3, "n": n}
    out = {
        "metric": "BA iterations/sec (W poses x F plane features)",
        "value": iters_per_s, "unit": "iter/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if (args.weak or args.features > 0 or world == 1) else "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[%d]: W=%d poses, %d plane features in total (%d per GPU), "
                               "%d pts per (feature,pose), full HIP accumulate + LDL^T LM solve"
                               % (2 if world == 1 else 3, W, F_total, Fg, args.pts),
                   "W": W, "features_per_gpu": Fg, "features_total": F_total, "form": "left",
                   "parallelism": ("features sharded x%d, one all-reduce of [tiles|blockdiag|r] per evaluation via %s"
                                   % (world, transport)) if multi else "single GPU"},
        "feature_iterations_per_sec": iters_per_s * F_total,     # size-normalised aggregate, comparable across N and F
        "kernel_ms_per_step": per_step,
        "roofline": roofline,
        "roofline_secondary": secondary,
        "natural_lm_run": {"what": "damping_iter from the noisy start to the reference's stop rule (u0=0.1, <=20 iterations), "
                                   "Hessian re-evaluated only after accepted steps; includes the pose upload/download",
                           "iterations": int(len(lg_nat)), "ms_total": t_nat * 1e3,
                           "iterations_per_sec": len(lg_nat) / t_nat, "final_residual": float(lg_nat[-1, 1])},
        "final_residual": float(lg[-1, 1]),
    }
    if world == 1 and not multi and not args.no_strong_ref and not args.no_cpu and args.features == 0 and W == 200:      # (--no-cpu: no extra legs at all)
        # the problem the N > 1 runs shard (BASELINE configs[3]: 200 000 features in total) on ONE GPU: the N = 1 point of the
        # strong-scaling curve (`value` above is configs[2], a 4x smaller problem, and not comparable with the N > 1 values)
        try:
            ctx.close()
            sc4 = scene.generate(args.seed, W, F_SHARDED_TOTAL, args.pts, mode=1)
            ctx4 = capi.Context(W, local_rank)
            ctx4.set_features(sc4.clusters, None, sc4.coeffs)
            ctx4.damping_iter(sc4.poses_init, form=0, u0=0.1, max_iter=3, force_hess=True, no_stop=True, reanchor=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k4 = 0
            for _ in range(2):
                ctx4.damping_iter(sc4.poses_init, form=0, u0=0.1, max_iter=20, force_hess=True, no_stop=True, reanchor=False)
                k4 += 20
            d4 = time.perf_counter() - t0
            out["strong_scaling_reference"] = {"what": "BASELINE configs[3] (W=200, 200 000 features) on one GPU: the N=1 point for the N>1 values",
                                               "features_total": F_SHARDED_TOTAL, "iterations_per_sec": k4 / d4, "ms_per_step": d4 / k4 * 1e3}
            ctx4.close()
            del sc4
            ctx = capi.Context(W, local_rank, capi.FLAG_TIMING)          # the CPU leg evaluates once on the device
            ctx.set_features(sc.clusters, None, sc.coeffs)
        except Exception as e:
            out["strong_scaling_reference"] = {"error": repr(e)}
    if world == 1 and not args.no_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline(sc, ctx, args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:   # the baseline leg must never take the GPU number down with it
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
