"""The arithmetic of BALM_FLAG_SYRK_INT8 (balm_amd/csrc/kernels_syrk_i8.hip) restated in numpy -- no GPU: the radix-254 digits of
k_i8_slice, the eleven digit products of k_syrk_i8 as exact integer matrix products, k_i8_pack's weights and scales.  What the GPU tests measure
on the device (tests/test_gpu_syrk_int8.py) must follow from this arithmetic: digits that fit an int8, int32 sums that cannot overflow at the
k-slice length the launcher chooses, and a product within the contract's error of the FP64 one."""
import numpy as np
import pytest

RADIX, DIGITS = 254.0, 4
PAIRS = [(a, b) for a in range(DIGITS) for b in range(DIGITS) if a + b <= 3 or (a, b) == (2, 2)]


def slice_rows(G):
    """k_i8_slice: one exponent per row (|x| / 2^e < 0.5), four round-to-nearest digits of radix 254"""
    m = np.abs(G).max(axis=1)
    e = np.where(m > 0, np.floor(np.log2(np.where(m > 0, m, 1.0))).astype(int) + 2, 0)
    r = G / np.exp2(e)[:, None]
    digits = []
    for _ in range(DIGITS):
        r = r * RADIX
        d = np.rint(r)
        r = r - d
        digits.append(d)
    return digits, np.where(m > 0, np.exp2(e), 0.0)


def sliced_product(G):
    digits, scale = slice_rows(G)
    n = G.shape[0]
    sets = np.zeros((5, n, n))
    for a, b in PAIRS:
        sets[4 if (a, b) == (2, 2) else a + b] += digits[a] @ digits[b].T          # exact: integers far below 2^53
    w = RADIX ** -(np.arange(5) + 2.0)
    return (sets * w[:, None, None]).sum(0) * scale[:, None] * scale[None, :], digits, sets


@pytest.mark.parametrize("seed,n,K,spread", [(1, 24, 500, 0.5), (2, 48, 20000, 1.5), (3, 7, 64, 3.0)])
def test_digits_fit_an_int8_and_the_product_is_within_the_contract(seed, n, K, spread):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, K)) * np.exp(spread * rng.standard_normal((n, K)))
    G[:, ::7] = 0.0
    G[n // 2] *= 1e-6                                        # a row six decades below its neighbours: its own exponent
    H, digits, sets = sliced_product(G)
    for d in digits:
        assert np.abs(d).max() <= 127                        # radix 254: round to nearest never needs +-128
    rowmax = np.abs(G).max(axis=1)
    rec = sum(d * RADIX ** -(a + 1.0) for a, d in enumerate(digits)) * np.where(rowmax > 0, np.exp2(np.floor(np.log2(rowmax)) + 2), 0)[:, None]
    assert np.abs(rec - G).max() <= 0.5 * RADIX ** -4.0 * 4.0 * rowmax.max() * 1.0001         # the remainder after four digits: half a unit of the last, 2^e <= 4 max
    # an entry's error: the dropped pairs a + b >= 4 (but (2, 2)) and the remainders, each <= K x 127^2 x 254^-6 x scale_i scale_j; measured far below
    bound = 8.0 * K * 127.0 ** 2 * RADIX ** -6.0 * np.outer(4.0 * rowmax, 4.0 * rowmax)
    assert (np.abs(H - G @ G.T) <= bound).all()
    # (relative to the largest entry of the product that is 2e-9 on these heavy-tailed rows -- largest |entry| 7 x the RMS -- and 1e-12 on the
    #  benchmark's factors, whose entries sit a median 3 bits under their row's largest: profiles/r06_int8_syrk_study.txt)
    assert np.array_equal(H, H.T)


def test_int32_sums_hold_a_k_slice_of_32704_columns():
    """the launcher cuts the columns into slices of at most 32 704 (kernels_syrk_i8.hip: syrk_i8_scratch_bytes): four pairs share the accumulator
    of a + b = 3, every digit product is at most 127^2"""
    assert 4 * 127 * 127 * 32704 < 2 ** 31
    assert 4 * 127 * 127 * (32704 + 64) >= 2 ** 31 - 4 * 127 * 127 * 64 * 40      # ... and not by a wide margin: the bound is the design's


def test_a_dropped_diagonal_pair_adds_coherently():
    """why (2, 2) is kept although a + b = 4: on the diagonal it is a sum of squares (tools/study_int8_syrk.py; profiles/r06_int8_syrk_study.txt)"""
    rng = np.random.default_rng(5)
    G = rng.standard_normal((16, 40000))
    digits, scale = slice_rows(G)
    diag22 = np.einsum("ik,ik->i", digits[2], digits[2])
    cross13 = np.einsum("ik,ik->i", digits[1], digits[3])
    assert (diag22 > 50 * np.abs(cross13)).all()
