"""CPU: the arithmetic identities the lattice-dosage kernel (rvtests_amd/csrc/suffstat_lat.hip.h) rests on, restated with
numpy integers — no GPU, no engine: the magic-number extraction of K = rint(g den) and its residual test, the two-digit
Gram identity with 32-bit wrap-around at the stated operand bound, and the 64-bit addition that decides (int)g' > 0."""
import numpy as np
import pytest

MAGIC = 6755399441055744.0          # 1.5 * 2^52


def _passes(e, K):
    """the kernel's acceptance test: |g den - K| <= K 2^-53 (lat_row: emax = max(|e| - r 2^-53) must stay <= 0)"""
    return np.abs(e) - np.asarray(K, dtype=np.float64) * 2.0 ** -53 <= 0.0


def _fma_residual(g, den, r):
    """fma(g, den, -r) for an integer-valued r: the exact product in integers of 2^-1074 would be overkill — g den is
    computed exactly with Python's fractions for the handful of values tested."""
    from fractions import Fraction
    return np.array([float(Fraction(x) * int(den) - Fraction(y)) for x, y in zip(g, r)])


@pytest.mark.parametrize("den", [1, 8, 100, 255, 1000, 2048])
def test_magic_number_gives_K_and_the_residual_separates_lattice_from_noise(den):
    rng = np.random.default_rng(den)
    K = rng.integers(0, 2 * den + 1, size=4000)
    g = K.astype(np.float64) / float(den)                    # the doubles strtod makes of "K / den"
    t = g * float(den) + MAGIC                                # (numpy rounds the product first: harmless for lattice points)
    low = t.view(np.uint64) & np.uint64(0xFFFFFFFF)
    assert np.array_equal(low.astype(np.int64), K)
    r = t - MAGIC
    assert np.array_equal(r, K.astype(np.float64))
    e = _fma_residual(g, den, r)
    assert _passes(e, K).all()                                # the double nearest to K / den always passes
    # EVERY lattice point of the denominator, not a sample
    Kall = np.arange(0, 2 * den + 1)
    gall = Kall.astype(np.float64) / float(den)
    eall = _fma_residual(gall, den, (gall * float(den) + MAGIC) - MAGIC)
    assert _passes(eall, Kall).all()
    # a value a few ulps beside a lattice point is NOT a lattice double: it must be handed back (round 3 accepted anything
    # within 2^-30 and silently snapped it to K / den in G'G) ...
    for ulps in (2, 3, 5, 100, 1 << 20):
        for sign in (-1, 1):
            x = gall[1:].copy()
            for _ in range(min(ulps, 5)):
                x = np.nextafter(x, np.inf if sign > 0 else -np.inf)
            if ulps > 5:
                x = gall[1:] * (1.0 + sign * ulps * 2.0 ** -52)
            tt = x * float(den) + MAGIC
            ee = _fma_residual(x, den, tt - MAGIC)
            assert not _passes(ee, tt - MAGIC).any(), (den, ulps, sign)
    # ... the upper neighbour of 1.0 (1 + 2^-52) as well; its lower neighbour (1 - 2^-53) sits exactly at the bound
    tt = np.nextafter(1.0, 2.0) * float(den) + MAGIC
    assert not _passes(_fma_residual([np.nextafter(1.0, 2.0)], den, [tt - MAGIC]), [tt - MAGIC])[0]
    # 0 passes only as 0.0
    tiny = np.array([5e-324, 1e-300, 2.0 ** -60])
    assert not _passes(_fma_residual(tiny, den, [0.0] * 3), [0.0] * 3).any()
    # ... and anything a mean imputation or another number of decimals produces does not pass
    if den >= 8:
        off = (K[:200].astype(np.float64) + rng.uniform(0.02, 0.98, 200)) / float(den)
        tt = off * float(den) + MAGIC
        ee = _fma_residual(off, den, tt - MAGIC)
        assert not _passes(ee, tt - MAGIC).any()


def test_range_test_on_the_high_dword():
    ok = np.array([0.0, 0.5, 1.0, 2.0, 2.0 + 2.0 ** -30])
    bad = np.array([-0.001, -1.0, 2.5, 4.0, np.inf, -np.inf, np.nan, 1e300])
    hi = lambda x: (x.view(np.uint64) >> np.uint64(32)).astype(np.uint64)
    assert (hi(ok) <= 0x40000000).all()
    assert (hi(bad) > 0x40000000).all()
    assert hi(np.array([-0.0]))[0] > 0x40000000                # (-0.0 is handed back: harmless, strtod never produces it for DS)


@pytest.mark.parametrize("den", [1000, 2048])
def test_two_digit_gram_is_exact_with_uint32_tiles_at_the_operand_bound(den):
    """lo = sum L'L and hi = sum [128 H'H + L'H + H'L] over 255 operands of 64 samples, accumulated modulo 2^32 as the int32
    matrix instruction does: K'K = lo + 128 hi exactly, for the worst case (every entry at its maximum) and a random one."""
    rng = np.random.default_rng(7)
    n, m = 64 * 255, 6
    for K in (np.full((n, m), 2 * den + 1, dtype=np.int64), rng.integers(0, 2 * den + 2, size=(n, m))):
        L, H = K & 127, K >> 7
        assert H.max() <= 32
        lo = np.zeros((m, m), dtype=np.uint32)
        hi = np.zeros((m, m), dtype=np.uint32)
        for o in range(255):
            sl = slice(64 * o, 64 * (o + 1))
            Lo, Ho = L[sl], H[sl]
            lo = (lo + (Lo.T @ Lo).astype(np.uint32)).astype(np.uint32)
            z = (Ho.T @ Ho).astype(np.uint32)
            hi = (hi + (z << np.uint32(7)) + (Lo.T @ Ho).astype(np.uint32) + (Ho.T @ Lo).astype(np.uint32)).astype(np.uint32)
        exact = K.T @ K
        assert np.array_equal(lo.astype(np.int64) + 128 * hi.astype(np.int64), exact)
        assert exact.max() < 2 ** 39                             # exact in fp64, and so is the sum over 10^3 wave-parts


def test_the_64_bit_addition_decides_the_burden_indicator():
    """(int)g' > 0: g >= 1.0 for an unflipped column, NOT g > 1.0 for a flipped one (g' = 2 - g) — bit 62 of bits + C."""
    one = np.float64(1.0)
    vals = np.array([0.0, 0.001, 0.5, np.nextafter(one, 0.0), 1.0, np.nextafter(one, 2.0), 1.001, 1.5, 2.0])
    bits = vals.view(np.uint64)
    unfl = ((bits + np.uint64(0x0010000000000000)) >> np.uint64(62)) & np.uint64(1)
    flip = (((bits + np.uint64(0x000FFFFFFFFFFFFF)) >> np.uint64(62)) & np.uint64(1)) ^ np.uint64(1)
    assert np.array_equal(unfl.astype(bool), vals.astype(np.int64) > 0)
    assert np.array_equal(flip.astype(bool), (2.0 - vals).astype(np.int64) > 0)


def test_digit_packing_by_byte_permutation():
    """w01 = K0 | K1 << 16, w23 likewise; low digits = bytes 0 and 2 of each word & 0x7f, high digits the same of w >> 7."""
    rng = np.random.default_rng(3)
    K = rng.integers(0, 4098, size=(1000, 4)).astype(np.uint32)
    w01 = K[:, 0] | (K[:, 1] << np.uint32(16))
    w23 = K[:, 2] | (K[:, 3] << np.uint32(16))

    def perm_06040200(s0, s1):      # v_perm_b32 D = {s1.b0, s1.b2, s0.b0, s0.b2}
        return (s1 & 0xFF) | (((s1 >> 16) & 0xFF) << 8) | ((s0 & 0xFF) << 16) | (((s0 >> 16) & 0xFF) << 24)
    b0 = perm_06040200(w23, w01) & np.uint32(0x7F7F7F7F)
    b1 = perm_06040200(w23 >> np.uint32(7), w01 >> np.uint32(7)) & np.uint32(0x7F7F7F7F)
    for l in range(4):
        assert np.array_equal((b0 >> np.uint32(8 * l)) & 0xFF, K[:, l] & 127)
        assert np.array_equal((b1 >> np.uint32(8 * l)) & 0xFF, K[:, l] >> 7)
