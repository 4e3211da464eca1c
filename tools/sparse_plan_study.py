#!/usr/bin/env python3
"""VERDICT r4 item 3, answered offline: can a POSE PERMUTATION bring the block-sparse SYRK plan's issued / algorithmic flops
(3.62 on the shipped window) down?  The plan is built on the host (balm_capi.hip: build_sparse_plan), so its cost can be
replayed exactly without a GPU: this script restates the planner (jobs of balm_create, (first, last, pattern) feature order,
16-feature chunks, a (job, chunk) item wherever the chunk touches the job's 80-row blocks), checks that the identity order
reproduces balm_work_model's figure, and then tries orders of the poses' row blocks:
  reverse Cuthill-McKee on the co-visibility graph (at several edge thresholds), the Fiedler vector of its Laplacian (plain and
  normalised), the poses sorted along each axis / the first principal axis of the trajectory, and a simulated annealing directly
  on sum_a nb_a (nb_a + 1) / 2  (nb_a = 80-row blocks feature a touches) -- a lower bound on what ANY order can reach here.
Input: a feature table [F, W, 10] (N in column 9).   python tools/sparse_plan_study.py [file.npz] > profiles/....txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILE, TM = 80, 5


def jobs_of(T):
    ng, jobs = T // 5, []
    for m in range(T):
        for i in range(m):
            jobs.append((0, i, m))
        if m >= TM * ng:
            jobs.append((0, m, m))
        elif m % TM == TM - 1:
            for v in (1, 2, 3):
                jobs.append((v, m - (TM - 1), m - (TM - 1)))
    return jobs


def touched_blocks(obs, pos):
    F, W = obs.shape
    T = (6 * W + TILE - 1) // TILE
    tb = np.zeros((F, T), bool)
    for i in range(W):
        for b in range((6 * pos[i]) // TILE, (6 * pos[i] + 5) // TILE + 1):
            tb[:, b] |= obs[:, i]
    return tb


def plan(obs, pos, C=16):
    """-> issued / algorithmic flops of the block-sparse plan, of the dense plan, mean blocks touched per feature"""
    F, W = obs.shape
    tb = touched_blocks(obs, pos)
    T = tb.shape[1]
    first = np.where(tb.any(1), tb.argmax(1), T)
    last = T - 1 - tb[:, ::-1].argmax(1)
    w = (1 << np.arange(T - 1, -1, -1, dtype=object))
    pat = np.array([int((tb[a] * w).sum()) for a in range(F)], dtype=object)
    order = sorted(range(F), key=lambda a: (first[a], last[a], -pat[a], a))
    jobs = jobs_of(T)
    total = 0
    for c in range(0, F, C):
        cm = np.concatenate([tb[order[c:c + C]].any(0), np.zeros(8, bool)])
        for (t, I, J) in jobs:
            if t == 0:
                need = cm[I] and cm[J]
            elif t == 1:
                need = cm[I] or cm[I + 1]
            elif t == 2:
                need = cm[I + 1] or cm[I + 2] or cm[I + 3]
            else:
                need = cm[I + 3] or cm[I + 4]
            total += bool(need)
    issued = total * (3 * C // 4) * 25 * 2048.0
    na = obs.sum(1).astype(float)
    alg = 216 * (na * (na + 1) / 2).sum()
    dense = len(jobs) * ((3 * F + 3) // 4) * 25 * 2048.0
    return issued / alg, dense / alg, tb.sum(1).mean()


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "oracle", "_ref", "realworld_features.npz")
    d = np.load(path)
    obs = d["clusters"][:, :, 9] != 0
    F, W = obs.shape
    ident = np.arange(W)

    def pos_of(order):
        pos = np.empty(W, int)
        pos[np.asarray(order)] = np.arange(W)
        return pos

    def show(name, pos):
        r, dn, nb = plan(obs, pos)
        hist = np.bincount(touched_blocks(obs, pos).sum(1), minlength=(6 * W + TILE - 1) // TILE + 1)
        print("%-34s issued/algorithmic %.3f  (dense plan %.2f)  blocks touched per feature: mean %.2f  histogram %s"
              % (name, r, dn, nb, " ".join(str(int(x)) for x in hist)))
        return r

    print("feature table %s: F = %d, W = %d, S = %d observations, block fill of the W x W pose pairs %.3f"
          % (os.path.basename(path), F, W, obs.sum(), ((obs.sum(1).astype(float) ** 2).sum()) / F / W / W))
    A = obs.T.astype(float) @ obs.astype(float)
    print("co-visibility graph: %.1f %% of the pose pairs share at least one feature" % (100 * (A > 0).mean()))
    show("identity (trajectory order)", ident)
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    for thr in (1, 5, 20, 50, 100):
        show("reverse Cuthill-McKee, edges >= %d" % thr, pos_of(reverse_cuthill_mckee(csr_matrix(A >= thr), symmetric_mode=True)))
    Wt = A.copy()
    np.fill_diagonal(Wt, 0)
    L = np.diag(Wt.sum(1)) - Wt
    show("Fiedler vector", pos_of(np.argsort(np.linalg.eigh(L)[1][:, 1])))
    Dm = np.diag(1 / np.sqrt(Wt.sum(1) + 1e-9))
    show("Fiedler vector, normalised Laplacian", pos_of(np.argsort(np.linalg.eigh(Dm @ L @ Dm)[1][:, 1])))
    if "poses" in d.files:
        p = d["poses"][:, 9:12]
        for ax in range(3):
            show("poses sorted along axis %d" % ax, pos_of(np.argsort(p[:, ax])))
        pc = p - p.mean(0)
        show("poses sorted along their principal axis", pos_of(np.argsort(pc @ np.linalg.svd(pc, full_matrices=False)[2][0])))
    # annealing on the tile count itself
    rng = np.random.default_rng(0)
    T = (6 * W + TILE - 1) // TILE

    def cost(pos):
        b = (6 * pos + 2) // TILE
        nb = np.zeros(F, int)
        for k in range(T):
            m = b == k
            if m.any():
                nb += obs[:, m].any(1)
        return (nb * (nb + 1) / 2).sum()

    pos = ident.copy()
    c = c0 = cost(pos)
    temp = 0.002 * c
    for it in range(int(os.environ.get("ANNEAL_STEPS", 20000))):
        i, j = rng.integers(0, W, 2)
        if (6 * pos[i] + 2) // TILE == (6 * pos[j] + 2) // TILE:
            continue
        pos[i], pos[j] = pos[j], pos[i]
        c2 = cost(pos)
        if c2 <= c or rng.random() < np.exp((c - c2) / temp):
            c = c2
        else:
            pos[i], pos[j] = pos[j], pos[i]
        temp *= 0.9997
    print("annealing on sum nb (nb + 1) / 2: %.0f -> %.0f tiles" % (c0, c))
    show("annealed order", pos)


if __name__ == "__main__":
    main()
