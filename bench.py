#!/usr/bin/env python3
"""bench.py -- BA iterations/s of the MI355X hot path on synthetic plane clouds.

  python bench.py --gpus N --steps K --warmup W
  N > 1, any of:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU, RCCL)
                  python bench.py --gpus N ...   (no launcher: re-executes itself under torch.distributed.run on 127.0.0.1; if that
                                                  launch fails, ONE process drives the N GPUs through balm_create_multi)

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
INT8_CEILING_TOPS = 3944.0  # MI355X_MICROARCH.md, matrix cores table: I8 >= 3944 TOPS dense (v_mfma_i32_16x16x64_i8 microbenchmark ceiling)
COPY_RATE_TBS = 6.29        # what the best plain copy kernel moves on the box (read + write; float4, one element per thread): tools/ubench_f64.hip,
                            # profiles/r04k_small_build_and_write_rate.txt (6.21 in r04c; the guide: 6.29).  Round 3 quoted 4.99 = its double4 grid-stride
                            # copy.  Read-only 6.25, write-only 6.95 TB/s with one element per thread (4.0-4.2 in the grid-stride form).
F_SINGLE = 50000            # BASELINE configs[2]
F_SHARDED_TOTAL = 200000    # BASELINE configs[3]
GOLDEN_SHARDED = os.path.join(ROOT, "tests", "golden", "lm_big_w200_f200000.npz")       # the reference's own run of configs[3]
N1_TRACE = os.path.join(ROOT, "profiles", "strong_scaling_n1_trace.json")                # this library's run of it on ONE GPU


def _pose_diff(a, b):
    """max rotation angle [rad] and translation distance [m] between two [W,12] pose arrays (R column-major, p)"""
    rot = tr = 0.0
    for x, y in zip(np.asarray(a), np.asarray(b)):
        D = x[:9].reshape(3, 3) @ y[:9].reshape(3, 3).T          # (R_a^T R_b)^T in column-major storage: same angle
        c = min(1.0, max(-1.0, (np.trace(D) - 1.0) * 0.5))
        sn = 0.5 * np.linalg.norm([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
        rot = max(rot, float(np.arctan2(sn, c)))
        tr = max(tr, float(np.linalg.norm(x[9:] - y[9:])))
    return rot, tr


def acceptance_check(ctx, poses_init, seed, W, F_total, pts):
    """The sharded problem's acceptance test inside the bench run itself (untimed): BALM2::damping_iter's own constants
    (bavoxel.hpp:1069-1166: u0 = 0.01, <= 10 iterations, >= 20 planes per pose, re-anchor) on configs[3], compared with
    (a) the REFERENCE's run of the same problem (tests/golden/lm_big_w200_f200000.npz, made by the reference's sources:
        same iteration count and accept sequence, (r1, r2) to the six decimals it prints, final poses <= 1e-5 rad / 1e-4 m),
    (b) this library's run of it on ONE GPU (profiles/strong_scaling_n1_trace.json, written by the N = 1 bench): r1, r2, q1
        to 1e-9 relative, poses to 1e-9 -- N ranks add the same payload in another order, nothing else may differ.
    Every rank calls this (it contains collectives); returns a dict for rank 0's JSON line."""
    out, lg = ctx.damping_iter(poses_init, form=0, u0=0.01, max_iter=10, min_planes=20)
    res = {"what": "configs[3] LM run with the reference's constants (u0=0.01, <=10 it., >=20 planes/pose), untimed",
           "iterations": int(len(lg)), "final_residual": float(lg[-1, 1]),
           "trace": [[float(x) for x in row[:7]] for row in lg], "ok": None}
    same_problem = (seed == 2024 and W == 200 and F_total == F_SHARDED_TOTAL and pts == 6)
    checks = []
    if same_problem and os.path.exists(GOLDEN_SHARDED):
        g = np.load(GOLDEN_SHARDED)
        rl, rp = g["lm_log_bavoxel"], g["lm_poses_bavoxel"]
        ok = len(rl) == len(lg) and bool(np.array_equal(rl[:, 6] > 0, lg[:, 6] > 0))
        if ok:
            ok = bool(np.all(np.abs(lg[:, :2] - rl[:, :2]) <= 1e-6 + 1e-8 * np.abs(rl[:, :2])) and np.all(np.abs(lg[:, 2] - rl[:, 2]) <= 1e-6))
        rot, tr = _pose_diff(out, rp)
        ok = ok and rot <= 1e-5 and tr <= 1e-4
        res["vs_reference"] = {"ok": ok, "max_rot_rad": rot, "max_trans_m": tr, "iterations_reference": int(len(rl)),
                               "fixture": "tests/golden/lm_big_w200_f200000.npz (bavoxel.hpp BALM2::damping_iter, 4 threads)"}
        checks.append(ok)
    if same_problem and os.path.exists(N1_TRACE):
        t1 = json.load(open(N1_TRACE))
        l1, p1 = np.array(t1["trace"]), np.array(t1["poses"])
        ok = len(l1) == len(lg) and bool(np.array_equal(l1[:, 6] > 0, lg[:, 6] > 0))
        if ok:
            ok = bool(np.allclose(lg[:, [0, 1, 5]], l1[:, [0, 1, 5]], rtol=1e-9, atol=0))
        rot, tr = _pose_diff(out, p1)
        ok = ok and rot <= 1e-9 and tr <= 1e-9
        res["vs_one_gpu"] = {"ok": ok, "max_rot_rad": rot, "max_trans_m": tr, "fixture": "profiles/strong_scaling_n1_trace.json"}
        checks.append(ok)
    res["ok"] = bool(all(checks)) if checks else None
    res["_poses"] = out
    return res


COMM_MODEL_MS = 0.03 + 0.07     # what the N = 1 line's prediction assumes per step: RCCL call overhead measured with a one-rank
                                # communicator (profiles/r04b_dist_overhead.txt) + link time of the 5.9 MB all-reduce


def one_gpu_same_problem(sc_full, W, device, steps=20):
    """The N = 1 point of THIS run's curve, measured by THIS run: the whole problem the N ranks are about to shard, on one GPU
    of the same box, same timed loop as the bench line (20 forced-Hessian LM steps from the noisy start), before any
    communicator exists in the process.  -> dict"""
    import torch
    from balm_amd import capi
    c = capi.Context(W, device, capi.FLAG_TIMING)
    c.set_features(sc_full.clusters, None, sc_full.coeffs)
    c.damping_iter(sc_full.poses_init, form=0, u0=0.1, max_iter=3, force_hess=True, no_stop=True, reanchor=False)
    c.reset_timing()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    done = 0
    while done < steps:
        m = min(20, steps - done)
        c.damping_iter(sc_full.poses_init, form=0, u0=0.1, max_iter=m, force_hess=True, no_stop=True, reanchor=False)
        done += m
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    tm = c.timing()
    c.close()
    return {"what": "the same %d-feature problem on ONE GPU of this box, timed by this run before any communicator existed "
                    "(%d forced-Hessian LM steps, as the bench line)" % (sc_full.F, steps),
            "features_total": int(sc_full.F), "steps": steps, "iterations_per_sec": steps / dt, "ms_per_step": dt / steps * 1e3,
            "kernel_ms_per_step": {k: v[0] / steps for k, v in tm.items()}}


def scaling_fields(ips_n, one_gpu, n_gpus):
    """top-level keys of an N > 1 line that let a reader compute nothing himself: speed-up and efficiency against the one-GPU
    run of the SAME problem measured in the same bench run (one_gpu = one_gpu_same_problem()'s dict, or None)"""
    if not one_gpu or not one_gpu.get("iterations_per_sec"):
        return {"speedup_vs_one_gpu_same_problem": None, "scaling_efficiency": None}
    sp = ips_n / one_gpu["iterations_per_sec"]
    return {"speedup_vs_one_gpu_same_problem": sp, "scaling_efficiency": sp / n_gpus}


def cpu_baseline(sc, ctx, target_seconds=20.0):
    """CPU baseline leg (rank 0, N=1), timed on the box's host cores on a bounded feature sample of the same
    workload and scaled by F/F_sample, plus one full (6W)^2 LDLT solve.  Build flags are the reference's own
    (-std=c++14 -O3, CMakeLists.txt:8-9).  Candidates:
      "reference"          oracle/_ref: the reference's bavoxel.hpp, BALM2::divide_thread_left with its 4 std::threads
                           (bavoxel.hpp:1027) -- the form BASELINE's real-world driver runs
      "reference_virtual"  oracle/_ref: the reference's benchmark_virtual.cpp, BALM2::left_evaluate_acc2 on ONE thread
                           (benchmark_virtual.cpp:413) -- the form BASELINE configs[0..3] name
      "reference_march_v3" "reference" rebuilt with -O3 -march=x86-64-v3 (the optional extra column of SURVEY 8d)
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
        # Two sample sizes, at least 32 and 64 features PER THREAD: every thread zeroes and the caller sums a private
        # (6W)^2 Hessian (as bavoxel.hpp:1042-1056 does), a cost that does not grow with the sample -- scaling one small
        # sample by F/F_sample multiplied it (round 2's "all cores" figure came out slower than 4 threads that way).
        # The full-size time is fixed + per_feature * F from the two points.
        f0 = min(64 * threads, F)
        te, tr = orc.time_sample(0, sc.clusters, None, sc.coeffs, sc.poses_init, f0, threads)
        f2 = max(min(F, 64 * threads), sample_size((te + tr) / f0, 0.35 * share))
        f1 = max(min(F, 32 * threads), f2 // 2)
        te1, tr1 = orc.time_sample(0, sc.clusters, None, sc.coeffs, sc.poses_init, f1, threads)
        te2, tr2 = orc.time_sample(0, sc.clusters, None, sc.coeffs, sc.poses_init, f2, threads)
        ts = orc.time_solve(H, g, 0.1)
        t1, t2 = te1 + tr1, te2 + tr2
        per = (t2 - t1) / (f2 - f1) if f2 > f1 else 0.0
        fixed = t2 - per * f2
        if not (per > 0 and fixed >= 0):            # noise beat the fit: plain scaling of the larger sample
            per, fixed = t2 / f2, 0.0
        cands[key] = dict(value=1.0 / (fixed + per * F + ts), f_sample=f2, f_samples=[f1, f2], threads=threads,
                          seconds_eval_sample=te2, seconds_resid_sample=tr2, seconds_samples=[t1, t2],
                          seconds_fixed=fixed, seconds_per_feature=per, seconds_solve=ts,
                          what="oracle left_evaluate_acc2 + evaluate_only_residual (fixed + per-feature cost fitted on two samples)")

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
    try:        # SURVEY 8d's optional extra column: the same sources with the host's vector ISA on (the reference's CMakeLists.txt only says -O3)
        from oracle import ref
        if ref.v3_available():
            te, tr = ref.time_sample_v3(sc.clusters, sc.coeffs, sc.poses_init, min(64, F))
            fr = min(sample_size((te + tr) / min(64, F), 0.5 * share), 6000)
            te, tr = ref.time_sample_v3(sc.clusters, sc.coeffs, sc.poses_init, fr)
            ts = ref.time_solve_v3(H, g, 0.1)
            cands["reference_march_v3"] = dict(value=1.0 / ((te + tr) * (F / fr) + ts), f_sample=fr, threads=4, seconds_eval_sample=te,
                                               seconds_resid_sample=tr, seconds_solve=ts,
                                               what="the reference's BALM2::divide_thread_left + evaluate_only_residual, built with -O3 "
                                                    "-march=x86-64-v3 (AVX2 + FMA; not the reference's own flags)")
    except Exception as e:
        cands["reference_march_v3_error"] = repr(e)
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
    eligible = [k for k in ("port", "reference", "reference_march_v3", "reference_virtual") if k in cands]
    kind_key = max(eligible, key=lambda k: cands[k]["value"])
    best = cands[kind_key]
    return {
        "value": best["value"], "unit": "iter/s", "cores": best["threads"],
        "kind": "reference" if kind_key.startswith("reference") else "port", "candidate": kind_key,
        "sample": "%s on the first %d of %d features (W=%d), scaled to F, + one full %dx%d LDLT solve; %d thread(s); "
                  "linear algebra = oracle/compat's stand-in Eigen (real Eigen is not in this image and would be faster)"
                  % (best["what"], best["f_sample"], F, W, 6 * W, 6 * W, best["threads"]),
        "host_cores": os.cpu_count(), "candidates": cands,
    }


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: run this same command line as N ranks under torch.distributed.run
    (what the driver's own N > 1 command does), pass rank 0's JSON line through.  Returns (exit code, the JSON line or None)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["BALM_BENCH_SELF_LAUNCHED"] = "1"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stdin=subprocess.DEVNULL, text=True,
                           timeout=float(os.environ.get("BALM_BENCH_LAUNCH_TIMEOUT", "1500")))
    except subprocess.TimeoutExpired:
        return 124, None
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    return p.returncode, line


def resolve_launch(args, world, argv, cuda_available, device_count):
    """How `--gpus N` is going to run.  Returns one of
         ("ranks", None)     started by a launcher (WORLD_SIZE set) or N = 1: this process is one rank
         ("self", None)      N > 1 without a launcher: re-execute under torch.distributed.run, fall back to "inproc"
         ("inproc", None)    N > 1 in ONE process through balm_create_multi (BALM_BENCH_INPROC=1 forces it)
         ("exit", (code, message))
    Only a missing GPU or fewer than N visible devices is an error; the launcher is never the caller's problem."""
    if not cuda_available:
        return "exit", (3, "bench.py: no GPU visible; the HIP path has no CPU fallback")
    if world > 1:
        if device_count < 1:
            return "exit", (3, "bench.py: no GPU visible; the HIP path has no CPU fallback")
        return "ranks", None
    if args.gpus <= 1:
        return "ranks", None
    if os.environ.get("BALM_BENCH_LOOPBACK") == "1":      # testing on a one-GPU box: the N shards of the one-process path on ONE device
        return "inproc", None
    if device_count < args.gpus:
        return "exit", (4, "bench.py: --gpus %d, but only %d GPU(s) are visible to this process" % (args.gpus, device_count))
    if os.environ.get("BALM_BENCH_INPROC") == "1":
        return "inproc", None
    return "self", None


def main(argv=None):
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
    ap.add_argument("--no-int8", action="store_true", help="N = 1: skip the extra leg that repeats the timed loop with BALM_SYRK=int8")
    ap.add_argument("--no-realworld", action="store_true", help="N = 1: skip the shipped-window end-to-end leg (datasets/realworld_w177.npz)")
    ap.add_argument("--no-one-gpu-ref", action="store_true",
                    help="N > 1: skip rank 0's run of the whole problem on one GPU (speedup_vs_one_gpu_same_problem / scaling_efficiency stay null)")
    ap.add_argument("--no-accept", action="store_true", help="N > 1: skip the untimed acceptance run against the reference's golden trace")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "hook"],
                    help="N > 1: RCCL inside the library (stream-ordered) or the torch.distributed hook")
    argv = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(argv)

    import torch
    from balm_amd import capi, dist as bdist, scene

    rank, local_rank, world = bdist.env_rank()
    mode, why = resolve_launch(args, world, argv, torch.cuda.is_available(), torch.cuda.device_count() if torch.cuda.is_available() else 0)
    if mode == "exit":
        print(why[1], file=sys.stderr)
        sys.exit(why[0])
    if mode == "self":
        rc, line = self_launch(args.gpus, argv)
        if line is not None and rc == 0:
            print(line, flush=True)
            return
        print("bench.py: launching %d ranks under torch.distributed.run failed (exit code %s); running the %d GPUs from this one "
              "process through balm_create_multi instead" % (args.gpus, rc, args.gpus), file=sys.stderr)
        mode = "inproc"
    inproc = mode == "inproc"
    n_gpus = args.gpus if inproc else world
    multi = world > 1 or os.environ.get("BALM_BENCH_FORCE_DIST") == "1"   # the latter: exercise the N>1 code path on one GPU
    torch.cuda.set_device(local_rank)
    W = args.win
    if args.features > 0:
        Fg = args.features
    elif n_gpus > 1 and not args.weak:
        Fg = F_SHARDED_TOTAL // n_gpus
    else:
        Fg = F_SINGLE
    # N > 1: rank 0 first runs the WHOLE problem alone on its GPU -- before the process group, before any RCCL communicator --
    # so that the line carries its own N = 1 point (the other ranks wait in the rendezvous meanwhile)
    one_gpu = None
    sc_full = None
    if n_gpus > 1 and rank == 0 and not args.no_one_gpu_ref:
        try:
            sc_full = scene.generate(args.seed, W, Fg * n_gpus, args.pts, mode=1, feature_offset=0)
            one_gpu = one_gpu_same_problem(sc_full, W, local_rank)
        except Exception as e:
            one_gpu = {"error": repr(e)}
        if not inproc:
            sc_full = None          # (a rank keeps only its shard)
    if multi:
        bdist.init_process_group("nccl")
    # every rank draws its own shard of one global scene: same trajectory and initial pose noise on all ranks
    # (engine(seed)), disjoint per-feature streams (feature_offset)
    # (one process driving all N GPUs draws the whole scene -- the same features -- and the context shards it)
    sc = sc_full if (inproc and sc_full is not None) else scene.generate(args.seed, W, Fg * n_gpus if inproc else Fg, args.pts, mode=1,
                                                                        feature_offset=0 if inproc else rank * Fg)
    sc_full = None

    loopback = inproc and os.environ.get("BALM_BENCH_LOOPBACK") == "1"
    ctx = (capi.Context(W, 0, capi.FLAG_TIMING | (capi.FLAG_LOOPBACK_SHARDS if loopback else 0), n_devices=n_gpus) if inproc
           else capi.Context(W, local_rank, capi.FLAG_TIMING))
    ctx.set_features(sc.clusters, None, sc.coeffs)
    transport = None
    if inproc:
        transport = "loopback shards on one device (testing)" if loopback else "rccl-in-library, one process (balm_create_multi)"
    if multi:
        if args.transport == "rccl":
            transport = bdist.install_rccl(ctx)
        else:
            bdist.install_allreduce(ctx)
            transport = "torch.distributed hook (host-synchronised)"

    def barrier():
        for d in (range(n_gpus) if (inproc and not loopback) else [local_rank]):
            torch.cuda.synchronize(d)
        if multi:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # the natural LM run first (untimed region of the contract; reported as its own figure): from the noisy start to
    # the reference's own stop rule, benchmark_virtual.cpp's constants
    H_first = ctx.evaluate(0, sc.poses_init)[0]    # (the process's first device work -- code objects, buffers -- is not the run's)
    barrier()
    t0 = time.perf_counter()
    poses_nat, lg_nat = ctx.damping_iter(sc.poses_init, form=0, u0=0.1, max_iter=20)
    barrier()
    t_nat = time.perf_counter() - t0

    def run_steps(k, ctx=ctx):
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
    comm = ctx.comm_info()
    accept = None
    if (multi or inproc) and not args.no_accept:
        accept = acceptance_check(ctx, sc.poses_init, args.seed, W, Fg * n_gpus, args.pts)
    if multi:
        import torch.distributed as dist
        if args.transport == "hook":
            ctx.set_allreduce(None)
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    F_total = Fg * n_gpus
    iters_per_s = args.steps / dt
    n = 6 * W
    per_step = {k: v[0] / args.steps for k, v in timing.items()}

    # One process driving all N devices: the work model is the WHOLE table's, a timer is ONE device's.  Every roofline figure below is
    # per device: a device's share of the work (the synthetic scene is dense: equal shards) over the mean of the devices' own launch
    # times (balm_get_shard_timing) -- round 5 divided the aggregate by device 0's time and read N x too high.
    shard_timings = [ctx.shard_timing(k) for k in range(n_gpus)] if inproc else [timing]
    share = 1.0 / n_gpus if inproc else 1.0

    def avg_s(key):
        per_dev = [t[key][0] / t[key][1] * 1e-3 for t in shard_timings if t[key][1]]
        return float(np.mean(per_dev)) if per_dev else None

    syrk_s = avg_s("syrk")          # HIP events around the kernel's launch on the library's stream
    achieved = share * wm["syrk_flops_algorithmic"] / syrk_s / 1e12 if syrk_s else None
    # HBM bytes per launch: NOT measured in this run (PMC counters need rocprofv3 around the process) -- read from the
    # committed summary of separate `rocprofv3 --pmc` passes of this same command; `traffic_source` names that run
    traffic, traffic_source = None, None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if pj.get("W") == W and pj.get("features_per_gpu") == Fg:
            traffic = pj["k_hessian_syrk"]["hbm_bytes_per_launch"]
            traffic_source = "profiles/pmc_traffic.json: %s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run)" % pj.get("run", "?")
    except Exception:
        pass
    roofline = {
        "kernel": "k_hessian_syrk", "bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": (achieved / FP64_PEAK_TFLOPS) if achieved else None, "traffic": traffic,
        "traffic_source": traffic_source,
        "dtype": "f64", "avg_launch_ms": syrk_s * 1e3 if syrk_s else None, "launches": timing["syrk"][1],
        "algorithmic_flops_per_launch": share * wm["syrk_flops_algorithmic"],
        "issued_flops_per_launch": share * wm["syrk_flops_issued"],
    }
    if inproc:
        roofline["per_device"] = "1/%d of the table's work over the mean launch time of the %d devices' own timers" % (n_gpus, n_gpus)
    # the other kernel classes of the step against THEIR rooflines (algorithmic bytes/flops of SURVEY 8d; S = observations)
    S = share * wm["S"]
    secondary = {}
    t = avg_s("moments")
    if t:      # K1 + K1b: 80 B per observation read (two evaluations per step: Hessian side and residual side)
        secondary["moments"] = {"bound": "hbm", "achieved": 80.0 * S / t / 1e12, "peak": HBM_PEAK_TBS, "unit": "TB/s",
                                "frac": 80.0 * S / t / 1e12 / HBM_PEAK_TBS, "frac_of_copy_rate": 80.0 * S / t / 1e12 / COPY_RATE_TBS,
                                "avg_launch_ms": t * 1e3}
    t = avg_s("factors")
    if t:      # K2: 80 B read + 144 B written per observation
        secondary["factors"] = {"bound": "hbm", "achieved": 224.0 * S / t / 1e12, "peak": HBM_PEAK_TBS, "unit": "TB/s",
                                "frac": 224.0 * S / t / 1e12 / HBM_PEAK_TBS, "frac_of_copy_rate": 224.0 * S / t / 1e12 / COPY_RATE_TBS,
                                "avg_launch_ms": t * 1e3}
    t = avg_s("solve")
    if t:      # blocked LDL^T: n^3/3 + 2 n^2 flops; a latency chain, priced against the FP64 peak for the record
        fl = n ** 3 / 3.0 + 2.0 * n * n
        secondary["solve"] = {"bound": "latency", "achieved": fl / t / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": fl / t / 1e12 / FP64_PEAK_TFLOPS, "avg_launch_ms": t * 1e3, "n": n}
    out = {
        "metric": "BA iterations/sec (W poses x F plane features)",
        "value": iters_per_s, "unit": "iter/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": None if n_gpus == 1 else ("weak" if (args.weak or args.features > 0) else "strong"),
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[%d]: W=%d poses, %d plane features in total (%d per GPU), "
                               "%d pts per (feature,pose), full HIP accumulate + LDL^T LM solve"
                               % (2 if n_gpus == 1 else 3, W, F_total, Fg, args.pts),
                   "W": W, "features_per_gpu": Fg, "features_total": F_total, "form": "left",
                   "parallelism": ("features sharded x%d, one all-reduce of [tiles|blockdiag|r] per evaluation via %s"
                                   % (n_gpus, transport)) if (multi or inproc) else "single GPU",
                   "launch": "one process, balm_create_multi" if inproc else
                             ("torch.distributed.run started by bench.py itself" if os.environ.get("BALM_BENCH_SELF_LAUNCHED") == "1" and world > 1
                              else ("torch.distributed.run" if world > 1 else "plain python"))},
        "feature_iterations_per_sec": iters_per_s * F_total,     # size-normalised aggregate, comparable across N and F
        "comm": {"transport": comm["transport"], "ranks_reported_by_transport": comm["ranks"], "world_size": n_gpus,
                 "payload_bytes_per_evaluation": comm["payload_doubles"] * 8,
                 "allreduce_ms_per_step": timing["comm"][0] / args.steps, "allreduces_per_step": timing["comm"][1] / args.steps,
                 "model_ms_per_step_assumed": COMM_MODEL_MS,      # what the N = 1 line's predicted curve charges per step: this run confirms or falsifies it
                 "note": "stream time of the all-reduces on rank 0 (HIP events around the RCCL calls on the library's stream): "
                         "includes waiting for the slowest rank"} if (multi or inproc) else None,
        "kernel_ms_per_step": per_step,
        "roofline": roofline,
        "roofline_secondary": secondary,
        "natural_lm_run": {"what": "damping_iter from the noisy start to the reference's stop rule (u0=0.1, <=20 iterations), "
                                   "Hessian re-evaluated only after accepted steps; includes the pose upload/download; warm (one evaluation ran before it)",
                           "iterations": int(len(lg_nat)), "ms_total": t_nat * 1e3,
                           "iterations_per_sec": len(lg_nat) / t_nat, "final_residual": float(lg_nat[-1, 1])},
        "final_residual": float(lg[-1, 1]),
    }
    # a fraction above 1 says "the timed kernel did not do the work it is credited with": refuse the line rather than print it
    fracs = [("roofline", roofline.get("frac"))] + [("roofline_secondary." + k, v.get("frac_of_copy_rate", v.get("frac"))) for k, v in secondary.items()]
    over = [name for name, f in fracs if f is not None and f > 1.0]
    if over:
        out["error"] = "roofline fraction above 1: " + ", ".join(over)
    if accept is not None:
        accept.pop("_poses", None)
        out["acceptance"] = accept
    if n_gpus > 1:
        out["one_gpu_same_problem"] = one_gpu
        out.update(scaling_fields(iters_per_s, one_gpu if one_gpu and "error" not in one_gpu else None, n_gpus))
    if n_gpus > 1 and out["scaling"] == "strong" and F_total == F_SHARDED_TOTAL and os.path.exists(N1_TRACE):
        # `value` at N = 1 is configs[2] (50 000 features), a 4x smaller problem than the one sharded here: the N = 1 point of THIS curve is the
        # same 200 000-feature problem on one GPU, measured by the N = 1 bench run (strong_scaling_reference) and committed with its LM trace
        try:
            t1 = json.load(open(N1_TRACE)).get("timing")
            if t1:
                out["same_problem_on_one_gpu_committed"] = {"iterations_per_sec": t1["iterations_per_sec"], "ms_per_step": t1["ms_per_step"],
                                                            "source": "profiles/strong_scaling_n1_trace.json (" + t1.get("run", "the N = 1 bench run") + "): "
                                                                      "another box's run -- a cross-check of one_gpu_same_problem, not the basis of scaling_efficiency",
                                                            "speedup_of_this_run": iters_per_s / t1["iterations_per_sec"]}
        except Exception:
            pass
    if n_gpus == 1 and not multi and not args.no_strong_ref and not args.no_cpu and args.features == 0 and W == 200:      # (--no-cpu: no extra legs at all)
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
                                               "features_total": F_SHARDED_TOTAL, "iterations_per_sec": k4 / d4, "ms_per_step": d4 / k4 * 1e3,
                                               }
            # the curve the measured N > 1 values are to be judged against: features shard, the assemble / solve /
            # pose update replicate, two RCCL calls per step (the 5.9 MB payload of an evaluation: ~0.07 ms of link time at
            # 7/8 x 2 x payload over 153 GB/s per link; the trial residual: 8 bytes).  comm = what those calls cost a step with a
            # ONE-rank communicator on one GPU (profiles/r04b_dist_overhead.txt: 4.286 vs 4.257 ms/step = 0.03 ms, the calls' own
            # stream time; round 3's 0.7 ms was the cooperative launch's cross-queue barriers, root-caused and removed there), plus
            # the wire time; an estimate until a multi-GPU node has measured it
            fixed_ms = per_step.get("solve", 0) + per_step.get("assemble", 0) + per_step.get("update", 0)
            t4 = d4 / k4 * 1e3
            comm_ms = COMM_MODEL_MS
            out["strong_scaling_reference"]["predicted"] = {
                "model": "T(N) = (T(1) - fixed) / N + fixed + comm; fixed = replicated solve + assemble + pose update of this run; "
                         "comm = one-rank RCCL call overhead measured on one GPU (0.03 ms/step, profiles/r04b_dist_overhead.txt) + link time of the 5.9 MB all-reduce (0.07 ms)",
                "fixed_ms": fixed_ms, "comm_ms_assumed": comm_ms,
                "ms_per_step": {str(N): (t4 - fixed_ms) / N + fixed_ms + comm_ms for N in (2, 4, 8)},
                "speedup_vs_one_gpu": {str(N): t4 / ((t4 - fixed_ms) / N + fixed_ms + comm_ms) for N in (2, 4, 8)}}
            # the same acceptance run the N > 1 benches make, here against the reference's golden trace only; its trace is
            # what they compare with to 1e-9 (written under gpurun_out/, committed as profiles/strong_scaling_n1_trace.json)
            if not args.no_accept:
                acc = acceptance_check(ctx4, sc4.poses_init, args.seed, W, F_SHARDED_TOTAL, args.pts)
                poses4 = acc.pop("_poses")
                acc.pop("vs_one_gpu", None)
                acc["ok"] = acc.get("vs_reference", {}).get("ok")
                out["strong_scaling_reference"]["acceptance"] = acc
                try:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    json.dump({"what": "balm_damping_iter (u0=0.01, <=10 it., >=20 planes) on configs[3], ONE GPU; rows: r1 r2 u v q q1 accepted",
                               "seed": args.seed, "W": W, "features_total": F_SHARDED_TOTAL, "pts": args.pts,
                               "timing": {"what": "the same %d timed LM steps as the bench line, on this one GPU" % k4,
                                          "ms_per_step": d4 / k4 * 1e3, "iterations_per_sec": k4 / d4},
                               "trace": acc["trace"], "poses": [[float(v) for v in row] for row in poses4]},
                              open(os.path.join(ROOT, "gpurun_out", "strong_scaling_n1_trace.json"), "w"))
                except Exception:
                    pass
            ctx4.close()
            del sc4
            ctx = capi.Context(W, local_rank, capi.FLAG_TIMING)          # the CPU leg evaluates once on the device
            ctx.set_features(sc.clusters, None, sc.coeffs)
        except Exception as e:
            out["strong_scaling_reference"] = {"error": repr(e)}
    if n_gpus == 1 and not multi and not args.no_cpu and not args.no_int8:
        # BALM_SYRK=int8 (opt-in): the same workload, the same timed loop, K3 on the INT8 matrix cores by error-free slicing.  An
        # EXTRA key: `value`, `dtype` and `roofline` above are the default FP64 path's.
        try:
            ctx.close()          # (contexts of one process share the device's CUs among their persistent solve kernels: a second live context
                                 #  shrinks the solve's helper grids -- 0.36 instead of 0.25 ms per step, measured -- so the leg runs alone)
            os.environ["BALM_SYRK"] = "int8"
            c8 = capi.Context(W, local_rank, capi.FLAG_TIMING)
            c8.set_features(sc.clusters, None, sc.coeffs)
            H8 = c8.evaluate(0, sc.poses_init)[0]
            poses8, lg8 = c8.damping_iter(sc.poses_init, form=0, u0=0.1, max_iter=20)
            if args.warmup > 0:
                run_steps(args.warmup, c8)
            c8.reset_timing()
            barrier()
            t0 = time.perf_counter()
            lg8s = run_steps(args.steps, c8)
            barrier()
            dt8 = time.perf_counter() - t0
            t8 = c8.timing()
            c8.close()
            rot8, tr8 = _pose_diff(poses8, poses_nat)
            out["int8_syrk"] = {
                "what": "the same %d timed LM iterations with BALM_SYRK=int8: Gt Gt^T as eleven exact int32 digit products on v_mfma_i32_16x16x64_i8 "
                        "(4 signed radix-254 digits per entry, one exponent per row), everything else unchanged; opt-in, not the default" % args.steps,
                "value": args.steps / dt8, "unit": "iter/s", "ms_per_step": dt8 / args.steps * 1e3,
                "speedup_vs_value": (args.steps / dt8) / iters_per_s,
                "dtype": "i8 digits x i8 digits -> i32 (exact), recombined in f64",
                "syrk_avg_launch_ms": t8["syrk"][0] / max(1, t8["syrk"][1]), "fp64_syrk_avg_launch_ms": syrk_s * 1e3 if syrk_s else None,
                # eleven digit products over the upper triangle: 11 x 2 x n(n+1)/2 x 3F int8 operations, over the WHOLE span (slicing + product + packing:
                # HIP events around the three kernels; k_syrk_i8 alone: profiles/r06z_int8_kernels_int8.txt) against the guide's INT8 MFMA ceiling
                "roofline": (lambda ops, sec: {"kernel": "k_i8_slice + k_syrk_i8 + k_i8_pack", "bound": "mfma", "dtype": "i8", "achieved": ops / sec / 1e12,
                                              "peak": INT8_CEILING_TOPS, "unit": "TOP/s", "frac": ops / sec / 1e12 / INT8_CEILING_TOPS,
                                              "algorithmic_ops_per_launch": ops})(11.0 * n * (n + 1) * 3.0 * F_total, t8["syrk"][0] / max(1, t8["syrk"][1]) * 1e-3),
                "kernel_ms_per_step": {k: v[0] / args.steps for k, v in t8.items()},
                "max_abs_H_difference_over_max_abs_diag_H": float(np.abs(H8 - H_first).max() / np.abs(np.diag(H_first)).max()),
                "natural_lm_run": {"iterations": int(len(lg8)), "iterations_fp64": int(len(lg_nat)), "final_residual": float(lg8[-1, 1]),
                                   "final_residual_fp64": float(lg_nat[-1, 1]), "max_pose_difference_to_fp64_run": {"rot_rad": rot8, "trans_m": tr8}},
                "final_residual": float(lg8s[-1, 1]),
            }
        except Exception as e:
            out["int8_syrk"] = {"error": repr(e)}
        finally:
            os.environ.pop("BALM_SYRK", None)
            ctx = capi.Context(W, local_rank, capi.FLAG_TIMING)
            ctx.set_features(sc.clusters, None, sc.coeffs)
    if n_gpus == 1 and not multi and not args.no_cpu and not args.no_realworld:
        # the path the reference ships data for (benchmark_realworld.cpp:183-218), end to end from host memory; an extra key,
        # outside the timed region of `value`
        from balm_amd import realworld as rw
        if os.path.exists(rw.SHIPPED_WINDOW_NPZ):
            try:
                ctx.close()
                out["realworld_end_to_end"] = rw.end_to_end(rw.SHIPPED_WINDOW_NPZ, local_rank)
                # ... and what the reference's C++ drivers get through include/balm_shim.hpp, scans held as pcl clouds (a fresh
                # process: its first call is the cold figure a one-shot driver pays)
                try:
                    cpp = rw.end_to_end_cpp(rw.SHIPPED_WINDOW_NPZ)
                    if cpp is not None:
                        late = rw.end_to_end_cpp(rw.SHIPPED_WINDOW_NPZ, reps=1, late=True)
                        cpp["declared_late"] = {k: late[k] for k in ("optimizer_object_declared", "ms_cold_create", "ms_cold_associate", "ms_cold_lm", "ms_cold_first_call")}
                    out["realworld_end_to_end"]["cpp_shim"] = cpp if cpp is not None else {
                        "skipped": "tools/bin/shim_realworld_e2e not built (tests/cpp/build_shim_driver.sh needs the reference's headers)"}
                except Exception as e:
                    out["realworld_end_to_end"]["cpp_shim"] = {"error": repr(e)}
                ctx = capi.Context(W, local_rank, capi.FLAG_TIMING)
                ctx.set_features(sc.clusters, None, sc.coeffs)
            except Exception as e:
                out["realworld_end_to_end"] = {"error": repr(e)}
        else:
            out["realworld_end_to_end"] = {"skipped": "datasets/realworld_w177.npz not present (tools/make_realworld_fixture.py writes it where the shipped data exists)"}
    if n_gpus == 1 and not args.no_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline(sc, ctx, args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:   # the baseline leg must never take the GPU number down with it
            out["cpu_baseline"] = {"error": repr(e)}
    # a sharded run must prove it really was N ranks: what the transport itself counts (ncclCommCount / the shard count)
    ranks_seen = out["comm"]["ranks_reported_by_transport"] if out.get("comm") else 1
    if out.get("error", "").startswith("roofline fraction"):
        print(json.dumps(out), flush=True)
        print("bench.py: " + out["error"], file=sys.stderr)
        sys.exit(6)
    if n_gpus > 1 and ranks_seen != n_gpus:
        out["error"] = "the transport reports %s ranks, the run was asked for %d" % (ranks_seen, n_gpus)
        print(json.dumps(out), flush=True)
        print("bench.py: " + out["error"], file=sys.stderr)
        sys.exit(5)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
