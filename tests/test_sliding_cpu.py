"""Host logic of balm_amd/sliding.py without a GPU: pose algebra and the window bookkeeping of SlidingWindowBA (which scan
gets which pose guess, what leaves the window when, what the trajectory holds), against a stand-in context that records
the calls the real one would receive."""
import numpy as np

from balm_amd.sliding import SlidingWindowBA, compose, inverse
from oracle import numpy_oracle as npo


def rand_pose(rng):
    R = npo.exp_so3(rng.standard_normal(3))
    return np.concatenate([R.T.reshape(9), rng.standard_normal(3)])


def test_pose_algebra():
    rng = np.random.default_rng(0)
    a, b, c = rand_pose(rng), rand_pose(rng), rand_pose(rng)
    ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=float)
    assert np.allclose(compose(a, inverse(a)), ident, atol=1e-14) and np.allclose(compose(inverse(a), a), ident, atol=1e-14)
    assert np.allclose(compose(compose(a, b), c), compose(a, compose(b, c)), atol=1e-13)
    x = rng.standard_normal(3)
    Ra, Rb = a[:9].reshape(3, 3).T, b[:9].reshape(3, 3).T
    assert np.allclose(compose(a, b)[:9].reshape(3, 3).T @ x + compose(a, b)[9:], Ra @ (Rb @ x + b[9:]) + a[9:], atol=1e-13)


class FakeContext:
    """records the window calls; its 'optimiser' shifts every pose by a known offset"""

    def __init__(self, W):
        self.W, self.calls, self.scans = W, [], 0

    def window_open(self, *a):
        self.calls.append(("open",) + a)

    def window_add_scan(self, xyz, pose):
        assert self.scans < self.W
        self.scans += 1
        self.calls.append(("add", int(xyz[0, 0]), pose.copy()))

    def window_features(self, want_features=True):
        self.calls.append(("features",))
        return 42, None

    def damping_iter(self, poses, **kw):
        assert poses.shape == (self.W, 12) and kw["reanchor"] is False
        out = poses.copy()
        out[:, 9:] += 0.5                       # "optimisation": every pose moves by (0.5, 0.5, 0.5)
        return out, np.zeros((3, 8))

    def window_marginalize(self, mg, poses):
        assert poses.shape == (self.W, 12)
        self.scans -= mg
        self.calls.append(("marg", mg, poses.copy()))


def test_sliding_window_bookkeeping():
    W, slide, total = 4, 2, 9
    ctx = FakeContext(W)
    ba = SlidingWindowBA(ctx, slide, voxel_size=1.5)
    assert ctx.calls[0][:2] == ("open", 1.5)
    ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=float)
    odom = [ident.copy() for _ in range(total)]
    for i in range(total):
        odom[i][9] = float(i)                   # odometry: one metre along x per scan
    results = []
    for i in range(total):
        r = ba.push(np.full((5, 3), i, np.float32), odom[i])
        if r is not None:
            results.append((i, r))
    # windows close at scans 3, 5, 7 (W = 4, slide 2)
    assert [i for i, _ in results] == [3, 5, 7] and all(r["F"] == 42 for _, r in results)
    adds = [c for c in ctx.calls if c[0] == "add"]
    assert [c[1] for c in adds] == list(range(total))
    # the first four scans go in with their odometry poses; scan 4 is chained onto the optimised pose of scan 3:
    # odometry increment (1, 0, 0) on top of (3 + 0.5, 0.5, 0.5)
    assert np.allclose(adds[3][2][9:], [3, 0, 0]) and np.allclose(adds[4][2][9:], [4.5, 0.5, 0.5])
    # scan 6 follows scan 5, which the second window moved again: 5.5 -> 6.0 (+1 along x)
    assert np.allclose(adds[6][2][9:], [7.0, 1.0, 1.0])
    margs = [c for c in ctx.calls if c[0] == "marg"]
    assert [m[1] for m in margs] == [slide] * 3 and np.allclose(margs[0][2][:, 9:], np.stack([o[9:] for o in odom[:4]]) + 0.5)
    traj = ba.trajectory()
    assert traj.shape == (total, 12)
    # y of every scan = 0.5 per optimisation it sat in, on top of what its pose guess inherited from its predecessor:
    # scans 0, 1: one window; 2, 3: two; 4, 5: guess at 0.5 + two windows; 6, 7: guess at 1.0 + one window so far; 8: guess at 1.5
    assert np.allclose(traj[:, 10], [0.5, 0.5, 1.0, 1.0, 1.5, 1.5, 1.5, 1.5, 1.5])


def test_a_window_that_cannot_be_optimised_still_slides():
    """damping_iter raising for 'no features' / the 20-planes precheck used to leave the device window full, so
    that every later push failed with 'window full'.  The window is reported as skipped and marginalised with the
    odometry-chained poses, as a stream of scans needs (the reference prints its message and stops, bavoxel.hpp:1079-1085)."""
    from balm_amd import capi

    class Starved(FakeContext):
        def __init__(self, W):
            super().__init__(W)
            self.windows = 0

        def window_features(self, want_features=True):
            self.windows += 1
            return (0 if self.windows == 2 else 42), None

        def damping_iter(self, poses, **kw):
            if self.windows == 1:
                raise capi.BalmError(capi.ERR_TOO_FEW_PLANES, "Initial error too large.")
            return super().damping_iter(poses, **kw)

    W, slide = 4, 2
    ctx = Starved(W)
    ba = SlidingWindowBA(ctx, slide)
    ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=float)
    out = []
    for i in range(8):
        p = ident.copy(); p[9] = i
        r = ba.push(np.full((5, 3), i, np.float32), p)
        if r is not None:
            out.append(r)
    assert [r["skipped"] is not None for r in out] == [True, True, False]
    assert "planes" in out[0]["skipped"] and out[1]["skipped"] == "no features"
    assert np.array_equal(out[0]["poses"], out[0]["poses_in"]) and len(out[0]["log"]) == 0
    margs = [c for c in ctx.calls if c[0] == "marg"]
    assert len(margs) == 3                      # every window slid, optimised or not
    assert np.allclose(out[2]["poses"][:, 10], 0.5)
