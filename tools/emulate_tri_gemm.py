#!/usr/bin/env python3
"""k_tri_gemm's data movement replayed in numpy (kernels_cov.hip, round 5): thread -> element of the 48 x 48 operand blocks -> LDS
(As[k][i], Bs[k][j], rows of 49 doubles) -> MFMA operands -> accumulator registers -> C, for the four stride patterns and triangular
modes of launch_congruence_inverse.  The kernel was written FROM this (no GPU in the loop while writing it); tests/test_capi_cpu.py runs it.
    python tools/emulate_tri_gemm.py [nA]        prints the largest deviation from the plain product per pattern"""
import sys

import numpy as np

NB, LD, T = 48, 49, 192


def run(nA, sa, sb, tri, rng):
    ldA = 2 * nA + NB
    size = max(nA * ldA, nA * nA) + 2 * ldA
    a, b = rng.standard_normal(size), rng.standard_normal(size)
    (sai, sak), (sbk, sbj) = sa, sb
    A = lambda i, k: a[i * sai + k * sak]
    B = lambda k, j: b[k * sbk + j * sbj]
    C = np.zeros(nA * nA)
    lanes = np.arange(64)
    l15, l4 = lanes & 15, lanes >> 4
    for bx in range(nA // NB):
        for by in range(nA // NB):
            i0, j0 = bx * NB, by * NB
            k0, k1 = 0, nA
            if tri == 1: k1 = min(nA, i0 + NB)
            elif tri == 2: k1 = min(nA, j0 + NB)
            elif tri == 3: k0 = i0
            elif tri == 4: k0 = j0
            acc = np.zeros((3, 3, 4, 64))                       # [wave][y][register][lane]
            for kb in range(k0, k1, NB):
                As, Bs = np.full(NB * LD, np.nan), np.full(NB * LD, np.nan)
                for t in range(T):                               # the 192 threads' twelve elements each
                    f, g0 = t % NB, t // NB
                    for m in range(12):
                        g = g0 + 4 * m
                        i, k = (f, g) if sai == 1 else (g, f)     # A block: the index along the unit stride is f
                        As[k * LD + i] = A(i0 + i, kb + k)
                        k2, j = (f, g) if sbk == 1 else (g, f)    # B block
                        Bs[k2 * LD + j] = B(kb + k2, j0 + j)
                for w in range(3):
                    for u in range(12):
                        av = As[(4 * u + l4) * LD + 16 * w + l15]
                        for y in range(3):
                            bv = Bs[(4 * u + l4) * LD + 16 * y + l15]
                            Am, Bm = np.zeros((16, 4)), np.zeros((4, 16))
                            Am[l15, l4], Bm[l4, l15] = av, bv    # v_mfma_f64_16x16x4_f64: A lane = (row, k), B lane = (col, k)
                            D = Am @ Bm
                            for e in range(4):
                                acc[w, y, e] += D[4 * e + l4, l15]   # D: row = (lane >> 4) + 4 register, col = lane & 15
            for w in range(3):
                for y in range(3):
                    for e in range(4):
                        C[(j0 + 16 * y + l15) * nA + i0 + 16 * w + l4 + 4 * e] = acc[w, y, e]
    Af = np.array([[A(i, k) for k in range(nA)] for i in range(nA)])
    Bf = np.array([[B(k, j) for j in range(nA)] for k in range(nA)])
    blk = lambda x: x // NB                                       # the operands are triangular; the kernel cuts k at block granularity
    I, K = np.meshgrid(np.arange(nA), np.arange(nA), indexing="ij")
    if tri == 1: Af = np.where(blk(K) <= blk(I), Af, 0)
    if tri == 3: Af = np.where(blk(K) >= blk(I), Af, 0)
    if tri == 2: Bf = np.where(blk(I) <= blk(K), Bf, 0)            # (I = k, K = j here)
    if tri == 4: Bf = np.where(blk(I) >= blk(K), Bf, 0)
    return np.abs(C.reshape(nA, nA).T - Af @ Bf).max()


def all_patterns(nA=96, seed=0):
    rng = np.random.default_rng(seed)
    ldA = 2 * nA + NB
    return [run(nA, (ldA, 1), (1, nA), 1, rng), run(nA, (1, nA), (1, ldA), 2, rng),
            run(nA, (1, ldA), (1, nA), 3, rng), run(nA, (1, nA), (ldA, 1), 4, rng)]


if __name__ == "__main__":
    print(all_patterns(int(sys.argv[1]) if len(sys.argv) > 1 else 144))
