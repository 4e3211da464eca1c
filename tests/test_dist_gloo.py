"""Multi-rank path on CPU: world_size-2 gloo processes.  The HIP kernels cannot run here, so the
per-rank evaluator is the oracle (tests may use it); what is covered is the product's sharding
logic (balm_amd.dist.partition_features), the payload exchange semantics (one sum all-reduce of
[H | g | r] per evaluation, one scalar per residual-only evaluation) and that a sharded LM loop with a
replicated solve reproduces the single-rank trajectory."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from balm_amd import dist as bdist
from oracle import orc
from util import make_scene, pose_errors, rel_err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, seed, W, F, pts, drop, out_q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    bdist.init_process_group("gloo")
    sc, _ = make_scene(seed, W, F, pts, drop)
    nobs = (sc.clusters[..., 9] > 0).sum(1)
    lo, hi = bdist.partition_features(nobs, world)[rank]
    cl, co = sc.clusters[lo:hi], sc.coeffs[lo:hi]

    def evaluate(poses):
        H, g, r = orc.evaluate(0, cl, None, co, poses)
        return bdist.allreduce_host_payload(H, g, r)

    def residual(poses):
        r = orc.only_residual(cl, None, co, poses)
        return bdist.allreduce_host_payload(np.zeros((0, 0)), np.zeros(0), r)[2]

    H, g, r = evaluate(sc.poses_init)
    # sharded LM loop, replicated solve (bavoxel.hpp:1104-1157 restated on the host)
    x = sc.poses_init.copy()
    u, v, calc = 0.01, 2.0, True
    trace = []
    for _ in range(10):
        if calc:
            Hc, gc, r1 = evaluate(x)
        dx, q1 = orc.solve_damped(Hc, gc, u)
        xt = orc.update_poses(0, x, dx)
        r2 = residual(xt)
        q = r1 - r2
        trace.append((r1, r2, u))
        if q > 0:
            x = xt
            q = q / q1; v = 2.0; q = 1 - (2 * q - 1) ** 3
            u *= max(1.0 / 3.0, q); calc = True
        else:
            u *= v; v *= 2; calc = False
        if abs(r1 - r2) / r1 < 1e-6:
            break
    x = orc.reanchor(x)
    if rank == 0:
        out_q.put((H, g, r, x, np.array(trace), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("seed,W,F,pts,drop", [(3, 12, 40, 10, 0.0), (4, 20, 61, 8, 0.4)])
def test_two_rank_sharded_evaluation_and_lm(seed, W, F, pts, drop):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, seed, W, F, pts, drop, q)) for r in range(2)]
    for p in procs:
        p.start()
    H, g, r, x, trace, shard = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc, _ = make_scene(seed, W, F, pts, drop)
    Ho, go, ro = orc.evaluate(0, sc.clusters, None, sc.coeffs, sc.poses_init)
    assert 0 < shard[1] < F
    assert rel_err(H, Ho) < 1e-12 and rel_err(g, go) < 1e-12 and abs(r - ro) / ro < 1e-13
    xo, lo = orc.damping_iter(0, sc.clusters, None, sc.coeffs, sc.poses_init, 0.01, 10)
    assert len(trace) == len(lo)
    assert np.allclose(trace[:, 0], lo[:, 0], rtol=1e-9) and np.allclose(trace[:, 1], lo[:, 1], rtol=1e-9)
    rot, tr = pose_errors(x, xo)
    assert rot.max() < 1e-9 and tr.max() < 1e-9


def test_partition_features_balances_syrk_cost():
    rng = np.random.default_rng(0)
    nobs = rng.integers(2, 178, size=2281)          # real-data-like co-visibility
    for world in (1, 2, 4, 8):
        parts = bdist.partition_features(nobs, world)
        assert parts[0][0] == 0 and parts[-1][1] == len(nobs)
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        assert all(hi > lo for lo, hi in parts)
        cost = np.array([(nobs[lo:hi] * (nobs[lo:hi] + 1) / 2).sum() for lo, hi in parts])
        assert cost.max() / cost.mean() < 1.05
    with pytest.raises(ValueError):
        bdist.partition_features([3, 3], 4)
    assert bdist.partition_features([5] * 8, 8) == [(i, i + 1) for i in range(8)]
