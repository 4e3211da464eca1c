import numpy as np
from balm_amd import capi, realworld as rw
d = np.load(rw.SHIPPED_WINDOW_NPZ)
xyz = np.ascontiguousarray(d["xyz"], dtype=np.float32); counts = d["counts"].astype(np.int64); poses = d["poses"]
offs = np.concatenate([[0], np.cumsum(counts)]); scans = [xyz[offs[i]:offs[i+1]] for i in range(len(counts))]
c = capi.Context(len(counts))
F, nr, (cl, co, lay, fix, pf) = c.associate_scans(scans, poses, 2.0, want_points=True)
n = xyz.shape[0]
print("features", F, "roots", nr, "points", n)
for L in range(3):
    print("layer", L, "features", int((lay == L).sum()), "points in them", int(co[lay == L].sum()), "= %.3f of all" % (co[lay == L].sum() / n))
print("points in no feature: %.3f" % ((pf < 0).sum() / n))
