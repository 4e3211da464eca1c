"""The sliding-window map against a committed fixture of the reference's octree used incrementally
(tests/golden/window_w8_mg2.npz, made by tests/golden/make_golden_window.py from OCTO_TREE_ROOT compiled out of
src/benchmark/bavoxel.hpp): the comparison that still runs where oracle/_ref is absent.

  -m "not gpu":  where oracle/_ref exists, the reference reproduces the fixture (the fixture is what it claims to be)
  -m gpu:        balm_window_* through the C ABI reproduces it: per-scan clusters bit for bit, fix clusters 1e-12
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden_window as mg  # noqa: E402

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "window_w8_mg2.npz")


def run(add, marg, snap):
    g = np.load(FIXTURE)
    seen = 0
    for step in mg.sequence():
        if step[0] == "add":
            add(step[1], step[2])
        elif step[0] == "marg":
            marg(step[1], step[2])
        else:
            cl, fix = mg.canon(*snap())
            ref_cl, ref_fix = g["cl_" + step[1]], g["fix_" + step[1]]
            assert cl.shape == ref_cl.shape, (step[1], cl.shape, ref_cl.shape)
            assert np.array_equal(cl, ref_cl), step[1]
            scale = np.abs(ref_fix).max(axis=1, keepdims=True) + 1e-300
            assert np.all(np.abs(fix - ref_fix) <= 1e-12 * scale) and np.array_equal(fix[:, 9], ref_fix[:, 9]), step[1]
            seen += 1
    assert seen == 1 + mg.SLIDES


def test_reference_reproduces_the_window_fixture():
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built")
    win = ref.Window(mg.W, voxel_size=1.0)
    run(win.add_scan, win.marginalize, lambda: win.features()[:2])
    win.close()


@pytest.mark.gpu
def test_window_map_reproduces_the_reference_fixture():
    from balm_amd import capi
    ctx = capi.Context(mg.W)
    ctx.window_open(voxel_size=1.0)

    def snap():
        F, (cl, co, layer, fix) = ctx.window_features()
        return cl, fix

    run(ctx.window_add_scan, ctx.window_marginalize, snap)
    ctx.close()
