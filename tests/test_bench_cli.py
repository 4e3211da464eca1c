"""bench.py's launch handling: `python bench.py --gpus N` must not depend on who started it."""
import argparse
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "BALM_BENCH_INPROC", "BALM_BENCH_LOOPBACK"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdin=subprocess.DEVNULL,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_resolve_launch_table():
    import bench
    ns = argparse.Namespace
    os.environ.pop("BALM_BENCH_INPROC", None)
    os.environ.pop("BALM_BENCH_LOOPBACK", None)
    # no GPU: the "no GPU" message, whatever N and whoever launched
    for world, gpus in ((1, 1), (1, 2), (2, 2), (1, 8)):
        mode, why = bench.resolve_launch(ns(gpus=gpus), world, [], False, 0)
        assert mode == "exit" and why[0] == 3 and "no GPU" in why[1]
    assert bench.resolve_launch(ns(gpus=1), 1, [], True, 1) == ("ranks", None)
    assert bench.resolve_launch(ns(gpus=8), 8, [], True, 8) == ("ranks", None)          # torch.distributed.run started us
    assert bench.resolve_launch(ns(gpus=2), 1, [], True, 8) == ("self", None)           # plain python: launch ourselves
    mode, why = bench.resolve_launch(ns(gpus=4), 1, [], True, 1)                        # fewer devices than asked for
    assert mode == "exit" and why[0] == 4 and "only 1 GPU" in why[1]
    os.environ["BALM_BENCH_INPROC"] = "1"
    try:
        assert bench.resolve_launch(ns(gpus=2), 1, [], True, 2) == ("inproc", None)
    finally:
        os.environ.pop("BALM_BENCH_INPROC")


def test_scaling_fields_of_an_n_gpu_line():
    """the N > 1 line's top-level speed-up / efficiency come from the one-GPU run of the same problem made by the same bench run"""
    import bench
    f = bench.scaling_fields(400.0, {"iterations_per_sec": 66.0}, 8)
    assert abs(f["speedup_vs_one_gpu_same_problem"] - 400.0 / 66.0) < 1e-12 and abs(f["scaling_efficiency"] - 400.0 / 66.0 / 8) < 1e-12
    assert bench.scaling_fields(400.0, None, 8) == {"speedup_vs_one_gpu_same_problem": None, "scaling_efficiency": None}
    assert bench.scaling_fields(400.0, {"error": "x"}, 8)["scaling_efficiency"] is None
    assert abs(bench.COMM_MODEL_MS - 0.10) < 1e-12


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="CPU-box behaviour")
def test_gpus_2_without_a_gpu_says_no_gpu():
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode == 3, p.stderr
    assert "no GPU visible" in p.stderr and "torch.distributed.run" not in p.stderr


@pytest.mark.gpu
def test_gpus_2_on_a_one_gpu_box_reaches_the_device_count_check():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with exactly one GPU")
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode == 4, p.stderr
    assert "only 1 GPU" in p.stderr


@pytest.mark.gpu
def test_one_process_fallback_runs_two_loopback_shards():
    """the balm_create_multi leg of bench.py (what runs if torch.distributed.run cannot start), on one device"""
    import json
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--features", "1500", "--win", "40", "--no-cpu", "--no-accept"],
             {"BALM_BENCH_LOOPBACK": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["features_total"] == 3000
    assert d["config"]["launch"] == "one process, balm_create_multi"
    assert d["comm"]["ranks_reported_by_transport"] == 2
    # the line is self-contained: its own one-GPU point of the same 3 000-feature problem, speed-up and efficiency from it
    one = d["one_gpu_same_problem"]
    assert one["features_total"] == 3000 and one["iterations_per_sec"] > 0
    assert abs(d["speedup_vs_one_gpu_same_problem"] - d["value"] / one["iterations_per_sec"]) < 1e-9
    assert abs(d["scaling_efficiency"] - d["speedup_vs_one_gpu_same_problem"] / 2) < 1e-12
    assert d["comm"]["allreduce_ms_per_step"] > 0 and d["comm"]["model_ms_per_step_assumed"] == 0.1
