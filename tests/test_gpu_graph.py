"""The LM iteration replayed as a hipGraph (opt-in, BALM_GRAPH=1; windows whose solve is the launch pair) must be
the same computation as the plain launch sequence: identical log and poses, bit for bit."""
import os

import numpy as np
import pytest

from balm_amd import capi
from util import make_scene

pytestmark = pytest.mark.gpu


def run(sc, fix, graph, **kw):
    if graph:
        os.environ["BALM_GRAPH"] = "1"
    else:
        os.environ.pop("BALM_GRAPH", None)
    try:
        c = capi.Context(sc.W)
        c.set_features(sc.clusters, fix, sc.coeffs)
        out = c.damping_iter(sc.poses_init, **kw)
        out2 = c.damping_iter(sc.poses_init, **kw)            # a second run on the same context reuses the graphs
        c.close()
    finally:
        os.environ.pop("BALM_GRAPH", None)
    assert np.array_equal(out[0], out2[0]) and np.array_equal(out[1], out2[1])
    return out


@pytest.mark.parametrize("W,F,drop", [(20, 20, 0.0), (33, 300, 0.4), (64, 2000, 0.0), (100, 800, 0.3)])
@pytest.mark.parametrize("form", [0, 1])
def test_graph_replay_is_the_same_run(W, F, drop, form):
    sc, fix = make_scene(W + F, W, F, 8, drop, with_fix=(form == 0 and drop > 0))
    for kw in (dict(form=form, u0=0.01, max_iter=10), dict(form=form, u0=0.1, max_iter=12, force_hess=True, no_stop=True),
               dict(form=form, u0=1e-6, max_iter=8)):          # tiny damping: rejected steps, the no-evaluate graph
        pa, la = run(sc, fix, True, **kw)
        pb, lb = run(sc, fix, False, **kw)
        assert np.array_equal(la, lb) and np.array_equal(pa, pb)


def test_graphs_survive_new_features_and_forms():
    sc, _ = make_scene(3, 24, 200, 8)
    sc2, _ = make_scene(4, 24, 350, 8, 0.2)
    c = capi.Context(24)
    os.environ.pop("BALM_GRAPH", None)
    ref = []
    for s_, form in ((sc, 0), (sc2, 0), (sc2, 1), (sc, 1)):
        c.set_features(s_.clusters, None, s_.coeffs)
        ref.append(c.damping_iter(s_.poses_init, form=form, u0=0.01, max_iter=10))
    os.environ["BALM_GRAPH"] = "1"
    for k, (s_, form) in enumerate(((sc, 0), (sc2, 0), (sc2, 1), (sc, 1))):
        c.set_features(s_.clusters, None, s_.coeffs)
        got = c.damping_iter(s_.poses_init, form=form, u0=0.01, max_iter=10)
        assert np.array_equal(got[0], ref[k][0]) and np.array_equal(got[1], ref[k][1])
    os.environ.pop("BALM_GRAPH", None)
    c.close()
