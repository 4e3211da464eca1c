#!/usr/bin/env python3
"""Where does the N > 1 code path's per-step overhead on ONE GPU come from (bench.py with BALM_BENCH_FORCE_DIST=1: 4.9 vs 4.2 ms)?
   python tools/exp_dist_overhead.py plain | torch_only | rccl_only | both
plain: no process group, no communicator; torch_only: torch.distributed (nccl) initialised, library without a communicator;
rccl_only: the library's own one-rank RCCL communicator, no torch.distributed; both: what bench.py does."""
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
print("%-10s %.3f ms/step | %s" % (mode, dt / K * 1e3, "  ".join("%s %.3f" % (k, v[0] / K) for k, v in t.items() if v[1])), flush=True)
ctx.close()
os._exit(0)
