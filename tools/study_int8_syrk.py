#!/usr/bin/env python3
"""could k_hessian_syrk (H -= Gt Gt^T, 2.17e11 FP64 flops, 76 % of the LM step, at the clock-limited FP64 MFMA rate) run
on the INT8 matrix cores by error-free slicing (Ozaki scheme) and still pass test_hip_full_size_hessian_against_the_reference_directly at its
UNCHANGED 1e-10?  CPU replay, no GPU minute.

  Gt (6W x 3F; W = 200, F = 50 000: 1200 x 150 000) is rebuilt in numpy from the golden scene (the factor columns of SURVEY.md 8(a) a4:
  sqrt(2 coe)/NN w, sqrt(2 coe/(lam_k - lam_0)) g_k).  A slicing writes every entry as  2^e * sum_{a < s} d_a 128^-(a+1),  d_a signed 7-bit
  digits, e = a shared exponent; slice products are exact in int32 (|d| <= 64: 64^2 * 2^16 columns < 2^31) and the scheme keeps a set of digit pairs (a, b).  Replayed with float64 GEMMs on the digit matrices (integers below 2^53: exact), per exponent-sharing choice:
      row           one exponent per row of Gt (1200)
      row x chunk   one exponent per row and per chunk of Kc feature columns (the int32 accumulation is cut into chunks of <= 2^16 columns
                    anyway); Kc = 65536 / 8192
  Reports the dynamic range of Gt's rows (bits below the row's largest entry, by quantile), then WHICH digit-by-digit products are needed:
  the error of candidate sets against the FP64 product in units of the test's scale (max |diag H| of the reference's evaluation), a greedy
  cheapest set that passes 1e-10 with a 3x margin, and what it prices the product at on the guide's INT8 ceiling (3944 TOPS,
  /opt/skills/guides/MI355X_MICROARCH.md).

    python tools/study_int8_syrk.py [--w 200 --f 50000]         (about ten minutes and 12 GB on 8 cores at full size)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from balm_amd import scene  # noqa: E402


def hat(v):
    z = np.zeros(v.shape[:-1])
    return np.stack([np.stack([z, -v[..., 2], v[..., 1]], -1), np.stack([v[..., 2], z, -v[..., 0]], -1), np.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def factor_columns(sc, poses, chunk=1000):
    """Gt [6W, 3F] of the left form (bavoxel.hpp:304-418 as SURVEY.md 8(a) a4 factors it), vectorised over (feature, pose)."""
    W, F = sc.W, sc.F
    R = poses[:, :9].reshape(W, 3, 3).transpose(0, 2, 1)
    p = poses[:, 9:]
    T = np.zeros((W, 4, 4)); T[:, :3, :3] = R; T[:, :3, 3] = p; T[:, 3, 3] = 1
    Gt = np.zeros((6 * W, 3 * F))
    diagB_scale = 0.0
    for a0 in range(0, F, chunk):
        cl = sc.clusters[a0:a0 + chunk]                       # [f, W, 10]
        f = cl.shape[0]
        Co = np.zeros((f, W, 4, 4))
        Co[..., 0, 0], Co[..., 0, 1], Co[..., 0, 2] = cl[..., 0], cl[..., 1], cl[..., 2]
        Co[..., 1, 1], Co[..., 1, 2], Co[..., 2, 2] = cl[..., 3], cl[..., 4], cl[..., 5]
        Co[..., 1, 0], Co[..., 2, 0], Co[..., 2, 1] = cl[..., 1], cl[..., 2], cl[..., 4]
        Co[..., :3, 3] = cl[..., 6:9]; Co[..., 3, :3] = cl[..., 6:9]; Co[..., 3, 3] = cl[..., 9]
        TC = np.einsum("wij,fwjk->fwik", T, Co)
        C = np.einsum("fwik,wlk->fil", TC, T)
        NN = C[:, 3, 3]
        Cn = C / NN[:, None, None]
        vbar = Cn[:, :3, 3]
        lam, U = np.linalg.eigh(Cn[:, :3, :3] - vbar[:, :, None] * vbar[:, None, :])       # [f,3], [f,3,3] (columns)
        coe = sc.coeffs[a0:a0 + chunk]
        Uk = np.zeros((3, f, 6, 4))
        for k in range(3):
            Uk[k, :, :3, :3] = hat(-U[:, :, k])
            Uk[k, :, 3:, 3] = U[:, :, k]
        tmp = np.broadcast_to(T[None, :, :3, :], (f, W, 3, 4)).copy()
        tmp[..., 3] -= vbar[:, None, :]
        M = np.einsum("fwij,fwkj->fwik", TC, tmp)                # [f,W,4,3]
        Mu = [np.einsum("fwik,fk->fwi", M, U[:, :, k]) for k in range(3)]      # M u_k: [f,W,4]
        g = [(np.einsum("fij,fwj->fwi", Uk[k], Mu[0]) + np.einsum("fij,fwj->fwi", Uk[0], Mu[k])) / NN[:, None, None] for k in (1, 2)]
        w = np.einsum("fij,fwj->fwi", Uk[0], TC[..., 3])
        obs = (cl[..., 9] > 0)[..., None]
        cols = [np.sqrt(2 * coe)[:, None, None] / NN[:, None, None] * w * obs,
                np.sqrt(2 * coe / (lam[:, 1] - lam[:, 0]))[:, None, None] * g[0] * obs,
                np.sqrt(2 * coe / (lam[:, 2] - lam[:, 0]))[:, None, None] * g[1] * obs]
        for k in range(3):
            Gt[:, 3 * a0 + k:3 * (a0 + f):3] = cols[k].reshape(f, 6 * W).T
    return Gt


def exponents(G, Kc):
    """shared exponents: e[i, c] with |G[i, k]| < 2^e for the columns k of chunk c (Kc = None: one chunk)"""
    n, K = G.shape
    Kc = K if Kc is None else Kc
    nch = (K + Kc - 1) // Kc
    e = np.zeros((n, nch), dtype=np.int32)
    for c in range(nch):
        m = np.abs(G[:, c * Kc:(c + 1) * Kc]).max(axis=1)
        e[:, c] = np.where(m > 0, np.floor(np.log2(np.maximum(m, 1e-300))).astype(np.int32) + 1, -1000)
    return e, Kc


def digits(G, e, Kc, s):
    """signed base-128 digits d_0..d_{s-1} of G / 2^e (|G / 2^e| < 1): G ~ 2^e sum_a d_a 128^-(a+1), |d_a| <= 64, round to nearest"""
    n, K = G.shape
    out = [np.zeros((n, K)) for _ in range(s)]
    for c in range(e.shape[1]):
        sl = slice(c * Kc, (c + 1) * Kc)
        r = np.ldexp(G[:, sl], -e[:, c][:, None])
        for a in range(s):
            r = r * 128.0
            d = np.rint(r)
            out[a][:, sl] = d
            r = r - d
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=200)
    ap.add_argument("--f", type=int, default=50000)
    ap.add_argument("--pts", type=int, default=6)
    ap.add_argument("--seed", type=int, default=None)
    a = ap.parse_args()
    gold = os.path.join(ROOT, "tests", "golden", "lm_big_w200_f50000.npz")
    g = dict(np.load(gold)) if (a.w, a.f) == (200, 50000) and os.path.exists(gold) else None
    seed = int(g["seed"]) if g is not None and a.seed is None else (a.seed or 1)
    pts = int(g["pts"]) if g is not None else a.pts
    t0 = time.time()
    sc = scene.generate(seed, a.w, a.f, pts, mode=1)
    G = factor_columns(sc, sc.poses_init)
    n, K = G.shape
    print("Gt %d x %d from scene seed %d (W = %d, F = %d, %d points per cluster), %.0f s" % (n, K, seed, a.w, a.f, pts, time.time() - t0))
    t0 = time.time()
    Href = G @ G.T
    t_gemm = time.time() - t0
    scale = float(np.abs(g["eval_diag"]).max()) if g is not None and "eval_diag" in g else float(np.abs(np.diag(Href)).max())
    print("FP64 product in %.1f s; max |Gt Gt^T| = %.4e, the test's scale (max |diag H| of the reference) = %.4e" % (t_gemm, np.abs(Href).max(), scale))
    if g is not None and "eval_diag" in g:
        # (H = blockdiag - Gt Gt^T: the diagonal of the product is of the order of the scale, so is its error budget)
        print("   ratio max |diag(Gt Gt^T)| / scale = %.3f" % (np.abs(np.diag(Href)).max() / scale))

    # ---- dynamic range of the rows --------------------------------------------------------------------------
    rmax = np.abs(G).max(axis=1)
    nz = G != 0
    bits = np.full(G.shape, np.nan)
    np.log2(rmax[:, None] / np.abs(G, where=nz, out=np.ones_like(G)), where=nz, out=bits)
    q = np.nanpercentile(bits, [1, 10, 50, 90, 99, 99.9], axis=1)
    print("\nbits below the row's largest entry, over the %d rows (rotation rows = 6i..6i+2, translation rows = 6i+3..6i+5):" % n)
    for name, sel in (("all rows", slice(None)), ("rotation rows", np.arange(n) % 6 < 3), ("translation rows", np.arange(n) % 6 >= 3)):
        qq = q[:, sel]
        print("   %-17s percentile of entries  1 %%: %5.1f   10 %%: %5.1f   50 %%: %5.1f   90 %%: %5.1f   99 %%: %5.1f   99.9 %%: %5.1f   (medians over rows)"
              % ((name,) + tuple(np.median(qq, axis=1))))
    print("   log2(largest row max / smallest row max) = %.1f bits; zero entries %.2f %%" % (np.log2(rmax.max() / rmax.min()), 100.0 * (1 - nz.mean())))
    h, edges = np.histogram(bits[nz], bins=np.arange(0, 72, 4))
    print("   histogram of all nonzero entries by bits below their row's maximum:")
    for lo, c in zip(edges[:-1], h):
        print("      %2d-%2d bits: %6.2f %%" % (lo, lo + 4, 100.0 * c / nz.sum()))
    del bits

    # ---- the slicings ---------------------------------------------------------------------------------------
    TOPS = 3944e12
    unit_ops = 108.0 * a.f * a.w * (a.w + 1)                  # one SYRK-shaped slice product (upper triangle), as DESIGN.md counts the FP64 flops
    smax = 6
    print("\nWhich slice products are needed?  Y_ab = A_a A_b^T (digit a against digit b), a <= b < %d.  A DROPPED product is an error: a cross" % smax)
    print("product (a != b) adds up with random signs over the %d columns, a diagonal one (a, a) is a sum of squares -- it adds up coherently on" % K)
    print("the diagonal of H, K / sqrt(K) = %.0f times worse.  (The textbook rule 'keep a + b <= s - 1' ignores that: it drops (s/2, s/2).)" % np.sqrt(K))
    print("Cost in SYRK units: (a, a) = 1, (a, b) = 2 (a full product: X + X^T needs both triangles of X).  Error in units of the test's scale, tolerance 1e-10.")
    for label, Kc in (("row", None), ("row x 65536", 65536), ("row x 8192", 8192)):
        t0 = time.time()
        e, kc = exponents(G, Kc)
        D = digits(G, e, kc, smax)
        Y = {}
        for c in range(e.shape[1]):
            sl = slice(c * kc, (c + 1) * kc)
            sc_row = np.ldexp(1.0, e[:, c])[:, None]
            S = [D[q][:, sl] * (128.0 ** -(q + 1)) * sc_row for q in range(smax)]       # digit matrices as values (exact: powers of two)
            for aa in range(smax):
                for bb in range(aa, smax):
                    X = S[aa] @ S[bb].T
                    Y[(aa, bb)] = Y.get((aa, bb), 0.0) + (X if aa == bb else X + X.T)
        del D

        def err(P):
            acc = np.zeros_like(Href)
            for pq in P:
                acc += Y[pq]
            return float(np.abs(acc - Href).max() / scale)

        def units(P):
            return sum(1 if p_ == q_ else 2 for p_, q_ in P)

        print("\n  exponent per %s  (%d chunk%s; %.0f s)" % (label, e.shape[1], "" if e.shape[1] == 1 else "s", time.time() - t0))
        for t in range(2, 6):
            P = [(p_, q_) for p_ in range(smax) for q_ in range(p_, smax) if p_ + q_ <= t]
            ep = err(P)
            print("     a + b <= %d                       %2d units  error %.2e  %s" % (t, units(P), ep, "PASS" if ep <= 1e-10 / 3 else "fail"))
            Pd = sorted(set(P + [(q_, q_) for q_ in range(smax) if 2 * q_ == t + 1]))
            if Pd != sorted(P):
                ed = err(Pd)
                print("     a + b <= %d and (%d, %d)            %2d units  error %.2e  %s" % (t, (t + 1) // 2, (t + 1) // 2, units(Pd), ed, "PASS" if ed <= 1e-10 / 3 else "fail"))
        # greedy: the product whose addition lowers the error most per unit, until the tolerance is met with a 3x margin
        P = [(0, 0)]
        cur = err(P)
        while cur > 1e-10 / 3 and len(P) < len(Y):
            best = None
            for pq in Y:
                if pq in P:
                    continue
                en = err(P + [pq])
                gain = (np.log(cur) - np.log(en)) / (1 if pq[0] == pq[1] else 2)
                if best is None or gain > best[0]:
                    best = (gain, pq, en)
            P.append(best[1]); cur = best[2]
        print("     greedy: %s" % " ".join("(%d,%d)" % pq for pq in P))
        print("     -> %d units, error %.2e; at the INT8 ceiling (%.0f TOPS): %.2f ms, at 60 %% of it: %.2f ms, at 40 %%: %.2f ms" %
              (units(P), cur, TOPS / 1e12, 1e3 * units(P) * unit_ops / TOPS, 1e3 * units(P) * unit_ops / TOPS / 0.6, 1e3 * units(P) * unit_ops / TOPS / 0.4), flush=True)
    print("\n(FP64 MFMA today: %.2f ms at the clock-limited 69 TF.  One SYRK unit costs %.0f us at the guide's 3944 TOPS.  Beside the products: the slicing of"
          " Gt -- 8 bytes read, one byte per digit written, fused into K2 or %.2f ms as a pass of its own at 5 TB/s -- and the int32 -> FP64 scaled sums of the"
          " products, about a dozen x %.1f MB: microseconds.)" % (1e3 * unit_ops / 69e12, 1e6 * unit_ops / TOPS, 1e3 * (8.0 + 5.0) * n * K / 5e12, n * n * 4 / 1e6))


if __name__ == "__main__":
    main()
