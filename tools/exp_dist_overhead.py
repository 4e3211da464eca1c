#!/usr/bin/env python3
"""Where does the N > 1 code path's per-step overhead on ONE GPU come from (bench.py with BALM_BENCH_FORCE_DIST=1: 4.9 vs 4.2 ms)?
   python tools/exp_dist_overhead.py plain | torch_only | rccl_only | both | rccl_other_ctx | rccl_destroyed [label]
plain: no process group, no communicator; torch_only: torch.distributed (nccl) initialised, library without a communicator;
rccl_only: the library's own one-rank RCCL communicator, no torch.distributed; both: what bench.py does;
rccl_other_ctx: ANOTHER context of the process holds the communicator, the measured one has none (is it the process or the stream?);
rccl_destroyed: a communicator was created and destroyed again before the measurement (state left behind, or live threads / queues?).
Also prints the process's thread count and the CPU time of its busiest threads (a spinning proxy thread shows here)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from balm_amd import capi, dist as bdist, scene

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
torch.cuda.set_device(0)
if mode in ("torch_only", "both"):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
    bdist.init_process_group("nccl")
W, F = 200, 50000
sc = scene.generate(2024, W, F, 6, mode=1)
def n_threads():
    return len(os.listdir("/proc/self/task"))
def thread_cpu():
    out = []
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read().rsplit(")", 1)[1].split()
            out.append(((int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK"), open("/proc/self/task/%s/comm" % t).read().strip()))
        except Exception:
            pass
    return sorted(out, reverse=True)[:5]
label = sys.argv[2] if len(sys.argv) > 2 else ""
th0 = n_threads()
other = None
if mode in ("rccl_other_ctx", "rccl_destroyed"):
    other = capi.Context(W, 0)
    other.comm_init_rank(1, 0, other.comm_unique_id())
    if mode == "rccl_destroyed":
        other.close(); other = None
ctx = capi.Context(W, 0, capi.FLAG_TIMING)
ctx.set_features(sc.clusters, None, sc.coeffs)
if mode == "both":
    bdist.install_rccl(ctx)
elif mode in ("rccl_only", "rccl_affinity"):
    before = sorted(os.sched_getaffinity(0))
    ctx.comm_init_rank(1, 0, ctx.comm_unique_id())
    after = sorted(os.sched_getaffinity(0))
    print("cpu affinity of the calling thread: %d cores before ncclCommInitRank, %d after%s" % (len(before), len(after), "" if before == after else " (CHANGED: %s...)" % after[:8]))
    if mode == "rccl_affinity":
        os.sched_setaffinity(0, before)
ctx.damping_iter(sc.poses_init, form=0, u0=0.1, max_iter=5, force_hess=True, no_stop=True, reanchor=False)
ctx.reset_timing()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 40
for _ in range(2):
    ctx.damping_iter(sc.poses_init, form=0, u0=0.1, max_iter=20, force_hess=True, no_stop=True, reanchor=False)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
t = ctx.timing()
print("threads: %d before, %d now; busiest (cpu s, name): %s" % (th0, n_threads(), thread_cpu()))
print("%-14s %-44s %.3f ms/step | %s" % (mode, label, dt / K * 1e3, "  ".join("%s %.3f" % (k, v[0] / K) for k, v in t.items() if v[1])), flush=True)
ctx.close()
os._exit(0)
