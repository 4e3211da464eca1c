"""CPU-side checks of the drop-in boundary: libbalm_hip.so loads and exports every symbol that
include/balm_hip.h declares; without a GPU the product path fails loudly (no CPU fallback)."""
import os
import re
import sys

import numpy as np
import pytest

from balm_amd import capi
from conftest import HAS_GPU, ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "balm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(balm_[a-z_]+)\s*\(", text)) - {"balm_allreduce_fn", "balm_fill_clusters_fn"})


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    syms = _declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(L, s), "libbalm_hip.so does not export %s" % s
    assert set(syms) == set(capi.EXPORTS)
    assert b"gfx950" in L.balm_version()


def test_product_package_never_touches_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, "balm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp", ".inc")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.lower(), "%s mentions the oracle" % os.path.join(dp, f)


def test_pinned_ring_upload_pipeline_on_a_fake_runtime(tmp_path):
    """balm_amd/csrc/host_stage.h (how the ABI's big host arrays reach HBM: a ring of pinned chunks filled by a pool of host
    threads beside the DMA) against tests/cpp/fakehip, whose hipMemcpyAsync is executed LATE by a thread: every byte arrives,
    every fill range exactly once and unit-aligned, no chunk refilled under a pending copy, the ring reused across calls."""
    import subprocess
    exe = str(tmp_path / "host_stage_test")
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(cpp, "fakehip"),
                           os.path.join(cpp, "host_stage_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "host_stage ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(HAS_GPU, reason="GPU present")
def test_fails_loudly_without_gpu():
    with pytest.raises(capi.BalmError):
        capi.Context(20)


def test_bad_window_is_rejected():
    with pytest.raises(capi.BalmError):
        capi.Context(0)
    with pytest.raises(capi.BalmError):
        capi.Context(100000)


def test_syrk_accumulators_stay_pinned():
    """the hand-pinned AGPR accumulators of k_hessian_syrk are only ever touched by the generated instructions
    (tools/check_syrk_agprs.py disassembles the built object): a compiler spill into an AGPR would corrupt the
    Hessian silently, and only the GPU parity tests would notice."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_syrk_agprs
    obj = os.path.join(ROOT, "balm_amd", "lib", "kernels_accum.o")
    if not os.path.exists(obj) or not os.path.exists(check_syrk_agprs.LLVM + "/llvm-objdump"):
        pytest.skip("object file or llvm-objdump not present")
    assert check_syrk_agprs.check(obj, verbose=False) == []


def test_persistent_solve_kernels_keep_their_coherent_stores_and_loads():
    """the hand-over protocol of k_ldl_chain / k_ldl_fused / k_ldl_backsolve rests on agent-scope atomics being lowered to
    write-through (sc1) stores, coherent loads and buffer_inv -- and on no buffer_wbl2 (tools/check_solve_sync.py
    disassembles the built object); a changed lowering would only show on a GPU, as rare wrong solutions"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_solve_sync
    obj = os.path.join(ROOT, "balm_amd", "lib", "kernels_solve.o")
    if not os.path.exists(obj) or not os.path.exists(check_solve_sync.LLVM + "/llvm-objdump"):
        pytest.skip("no built object / no llvm-objdump")
    assert check_solve_sync.check(obj, verbose=False) == []


@pytest.mark.parametrize("P,NH", [(31, 223), (38, 189), (63, 191), (75, 179), (100, 154), (5, 3), (64, 8)])
def test_macro_tile_ownership_plan_is_a_balanced_partition(P, NH):
    """k_ldl_chain's helpers on 2 x 2 macro-tiles (csrc/kernels_chain.inc: chain_macro_plan, host code): every macro-tile of
    the tall matrix [A ; rhs] below the first column pair has exactly one owner, a helper's list ascends by column (its lowest
    ready lane is then the most urgent job), nobody holds more than the 64 a wavefront can schedule, and the work -- panels x
    (2 + live tiles), the cost the plan balances -- is even: dealt out boustrophedon the busiest helper did 354 tile updates
    at 63 panels and the idlest 92 (profiles/r03z_chain_helpers.txt)"""
    from balm_amd import capi
    want = {(r0, j0) for j0 in range(2, P, 2) for r0 in range(j0, P + 1, 2)}
    try:
        plan = capi.chain_macro_plan(P, NH)
    except capi.BalmError:
        assert len(want) > 64 * NH
        return
    got = [t for row in plan for t in row]
    assert len(got) == len(set(got)) and set(got) == want
    assert all(len(row) <= 64 for row in plan)
    for row in plan:
        assert [(j0, r0) for r0, j0 in row] == sorted((j0, r0) for r0, j0 in row)

    def cost(r0, j0):
        w = 0
        for q in range(P):
            n_on = sum(1 for dr in (0, 1) for dc in (0, 1)
                       if r0 + dr <= P and j0 + dc < P and r0 + dr >= j0 + dc and q <= ((j0 + dc - 3) if r0 + dr == j0 + dc else (j0 + dc - 2)))
            w += 2 + n_on if n_on else 0
        return w
    loads = [sum(cost(*t) for t in row) for row in plan]
    heaviest = max(cost(*t) for t in want)
    if len(want) >= NH:
        assert min(len(row) for row in plan) >= 1
    assert max(loads) <= sum(loads) / NH + heaviest          # longest-processing-time-first's bound
    if P == 63:
        assert max(loads) <= 1.12 * sum(loads) / NH


def test_every_environment_switch_is_documented_and_exercised():
    """a switch of the library that no test executes selects unrun code.  Every getenv("BALM_...") in
    balm_amd/csrc has a row in INTEGRATION.md's table and a hit in tests/ -- and there are no more of them than the review allowed (18)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for f in glob.glob(os.path.join(root, "balm_amd", "csrc", "*")):
        if os.path.isfile(f):
            names |= set(re.findall(r'getenv\("(BALM_[A-Z0-9_]+)"\)', open(f, errors="ignore").read()))
    assert 10 <= len(names) <= 18, sorted(names)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    tests = "".join(open(f).read() for f in glob.glob(os.path.join(root, "tests", "*.py")) if not f.endswith("test_capi_cpu.py"))
    for n in sorted(names):
        assert n in doc, "%s is not in INTEGRATION.md's switch table" % n
        assert n in tests, "%s is not exercised by any test" % n


def test_tri_gemm_index_arithmetic_on_the_numpy_replay():
    """k_tri_gemm (the covariance congruence's triangular products through LDS) was written from a numpy replay of its data movement --
    thread -> block element -> LDS -> MFMA operand -> accumulator -> C; the replay must reproduce the plain product for the four stride
    patterns and triangular modes it is launched with (the GPU tests then compare the kernel itself with the reference's covariance)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import emulate_tri_gemm
    assert max(emulate_tri_gemm.all_patterns(96)) < 1e-12

