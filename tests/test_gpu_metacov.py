"""GPU parity of the MetaCov covariance band (rvt_cov_block, through the C ABI) against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

import orc
import synth
from test_metacov_cpu import make_case

pytestmark = pytest.mark.gpu

REL = 1e-9   # of the largest covariance of the block (values are differences of O(N) sums)


def run_case(engine_factory, N, V, d, binary, seed, window):
    G, chrom, pos, X, y = make_case(N, V, d, binary, seed)
    if binary:
        rc, beta, p, v = orc.fit_logistic(X, y)
        assert rc == 0
        res, s2 = y - p, 1.0
    else:
        rc, beta, pred, res, s2 = orc.fit_linear(X, y)
        assert rc == 0
        v = np.full(N, s2)
    eng = engine_factory()
    eng.set_null(binary, X, res, v, s2)
    ptr = eng.upload_block(G)
    cov, xz, zz, poly = eng.cov_block(ptr, V)
    rc, kept, ocov, row_end, oxz, ozz = orc.metacov(G, chrom, pos, X, y, binary, window)
    assert rc == 0
    assert (poly == kept).all()
    m = ~np.isnan(ocov)                        # the pairs the reference prints
    assert m.sum() > V
    scale = np.abs(ocov[m]).max()
    assert np.abs(cov[m] - ocov[m]).max() < REL * scale
    kk = kept.astype(bool)
    assert np.allclose(xz[kk], oxz[kk], rtol=1e-9, atol=1e-9 * max(np.abs(oxz[kk]).max(), 1.0))
    assert np.allclose(zz, ozz, rtol=1e-9, atol=1e-9 * max(np.abs(ozz).max(), 1.0))
    eng.free_block(ptr)
    return eng


@pytest.fixture
def engine_factory():
    import rvtests_amd
    made = []

    def make():
        e = rvtests_amd.Engine(0)
        made.append(e)
        return e
    yield make
    for e in made:
        e.close()


@pytest.mark.parametrize("binary", [0, 1])
@pytest.mark.parametrize("V,d", [(23, 3), (64, 1), (97, 2), (200, 3)])
def test_cov_block_matches_oracle(engine_factory, binary, V, d):
    run_case(engine_factory, 1500, V, d, binary, 100 + V + d + 13 * binary, window=3000)


def test_cov_block_ring_moves(engine_factory):
    """upload_columns / move_columns (the adapter's device ring): a block assembled column by column and compacted
    gives the same band as the block uploaded at once."""
    N, V, d = 1200, 40, 2
    G, chrom, pos, X, y = make_case(N, V, d, 0, 77)
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    eng = engine_factory()
    eng.set_null(0, X, res, np.full(N, s2), s2)
    whole = eng.upload_block(G)
    ref = eng.cov_block(whole, V)[0]
    ring = eng.upload_block(np.zeros((N, V + 8)))
    for j in range(V):
        eng.upload_columns(ring, j + 8, G[:, j])
    eng.move_columns(ring, 0, 8, V)
    got = eng.cov_block(ring, V)[0]
    iu = np.triu_indices(V)
    assert np.array_equal(got[iu], ref[iu])


def test_column_cache_of_uploaded_columns(engine_factory, monkeypatch):
    """rvt_block_upload_columns leaves with every hard-call column what MetaCov's column pass would compute for it (int8 copy,
    sum, polymorphic flag, row of T = G'X), so that a covariance call on such a block starts at the integer product.  The
    numbers must be those of the same block uploaded at once — bit for bit —; the cache travels with moved and copied columns,
    is not used after the null model changed, and a dosage column switches the block to the general path."""
    N, V, d = 9000, 150, 3                                   # (N > 2 x 4096: several row slices)
    G, chrom, pos, X, y = make_case(N, V, d, 0, 31337)
    G = np.rint(G)
    G[:, 11] = 0.0                                            # monomorphic
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    eng = engine_factory()
    eng.set_null(0, X, res, np.full(N, s2), s2)
    whole = eng.upload_block(G)
    assert eng.classify_block(whole, V)
    ref = eng.cov_block(whole, V)
    ring = eng.alloc_block(V + 20)
    for j0 in range(0, V, 37):
        eng.upload_columns(ring, 20 + j0, G[:, j0:j0 + 37])
    eng.move_columns(ring, 0, 20, V)                          # overlapping forward move: the cache moves with the columns
    got = eng.cov_block(ring, V)
    iu = np.triu_indices(V)
    assert np.array_equal(got[0][iu], ref[0][iu]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[3], ref[3])
    monkeypatch.setenv("RVT_METACOV_NO_CACHE", "1")           # the same call without the cache
    got2 = eng.cov_block(ring, V)
    monkeypatch.delenv("RVT_METACOV_NO_CACHE")
    assert np.array_equal(got2[0][iu], ref[0][iu]) and np.array_equal(got2[1], ref[1])
    r1 = eng.cov_rect(ring, 7, 64, 120)                       # a rectangle inside the cached block
    r0 = eng.cov_rect(whole, 7, 64, 120)
    for h in range(64):
        assert np.array_equal(r1[0][h, h:], r0[0][h, h:], equal_nan=True)
    other = eng.alloc_block(V)                                # device copies hand the cache on
    eng._check(eng.L.rvt_block_copy_columns(eng.ctx, C.c_void_p(other), 0, C.c_void_p(ring), 0, V))
    got3 = eng.cov_block(other, V)
    assert np.array_equal(got3[0][iu], ref[0][iu]) and np.array_equal(got3[1], ref[1])
    # another null model: the cached rows of T belong to the old one and must not be used
    X2 = X.copy()
    X2[:, 1] = X[:, 1] ** 2
    rc, beta2, pred2, res2, s22 = orc.fit_linear(X2, y)
    eng.set_null(0, X2, res2, np.full(N, s22), s22)
    got4 = eng.cov_block(ring, V)
    ref4 = eng.cov_block(whole, V)
    assert np.array_equal(got4[0][iu], ref4[0][iu]) and np.array_equal(got4[1], ref4[1])
    assert not np.array_equal(ref4[1], ref[1])
    # a dosage in one column: its flag says so and the block takes the general path (same numbers as the block uploaded at once)
    G2 = G.copy()
    G2[3, 40] = 0.25
    eng.upload_columns(ring, 40, G2[:, 40])
    whole2 = eng.upload_block(G2)
    got5 = eng.cov_block(ring, V)
    ref5 = eng.cov_block(whole2, V)
    scale = np.nanmax(np.abs(ref5[0][iu]))
    assert np.abs(got5[0][iu] - ref5[0][iu]).max() <= 1e-12 * scale and np.array_equal(got5[3], ref5[3])


@pytest.mark.parametrize("binary", [0, 1])
def test_cov_rect_matches_block(engine_factory, binary):
    """The heads-by-window rectangle (two GEMMs; for windows wider than one block) gives the same band as the
    symmetric block kernel, and the same numbers as the oracle."""
    N, V, d = 1500, 90, 3
    G, chrom, pos, X, y = make_case(N, V, d, binary, 321 + binary)
    if binary:
        rc, beta, p, v = orc.fit_logistic(X, y)
        res, s2 = y - p, 1.0
    else:
        rc, beta, pred, res, s2 = orc.fit_linear(X, y)
        v = np.full(N, s2)
    eng = engine_factory()
    eng.set_null(binary, X, res, v, s2)
    ptr = eng.upload_block(G)
    cov, xz, zz, poly = eng.cov_block(ptr, V)
    col0, H, W = 7, 20, 70
    rcov, rxz, rzz, rpoly = eng.cov_rect(ptr, col0, H, W)
    assert (rpoly == poly[col0:col0 + W]).all()
    scale = np.nanmax(np.abs(cov[np.triu_indices(V)]))
    for h in range(H):
        for j in range(h, W):
            assert abs(rcov[h, j] - cov[col0 + h, col0 + j]) <= 1e-9 * scale
    assert np.allclose(rxz, xz[col0:col0 + W], rtol=1e-9, atol=1e-9 * max(np.abs(xz).max(), 1.0))
    assert np.allclose(rzz, zz, rtol=1e-12, atol=0)


def test_cov_rect_hard_call_fast_path(engine_factory, monkeypatch):
    """Heads against a wider window of a hard-call block (what the adapter's ring hands over once it holds more than 1 024
    columns) take the exact int8 product too (round 5): the rows equal the symmetric block's BIT FOR BIT (the products are
    integers either way), with a monomorphic column and a non-zero column offset; a dosage inside the window falls back."""
    N, V, d = 2100, 260, 2
    G, chrom, pos, X, y = make_case(N, V, d, 0, 909)
    G = np.rint(G)
    G[:, 31] = 1.0
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    eng = engine_factory()
    eng.set_null(0, X, res, np.full(N, s2), s2)
    ptr = eng.upload_block(G)
    assert eng.classify_block(ptr, V)
    cov, xz, zz, poly = eng.cov_block(ptr, V)
    for col0, H, W in ((0, 100, 260), (9, 64, 200), (20, 1, 64)):
        eng.set_profiling(True)
        eng.timing(reset=True)
        rcov, rxz, rzz, rpoly = eng.cov_rect(ptr, col0, H, W)
        assert (rpoly == poly[col0:col0 + W]).all()
        for h in range(H):
            assert np.array_equal(rcov[h, h:], cov[col0 + h, col0 + h:col0 + W], equal_nan=True)
        assert np.array_equal(rxz, xz[col0:col0 + W])
    # a dosage in the window: the int8 pass reports it and the call is computed again the general way
    G2 = G.copy()
    G2[5, 100] = 0.37
    ptr2 = eng.upload_block(G2)
    rcov, rxz, rzz, rpoly = eng.cov_rect(ptr2, 9, 64, 200)
    rc, kept, ocov, row_end, oxz, ozz = orc.metacov(G2, chrom, pos, X, y, 0, 10 ** 7)
    scale = np.nanmax(np.abs(ocov))
    for h in range(64):
        for j in range(h, 200):
            if kept[9 + h] and kept[9 + j] and not np.isnan(ocov[9 + h, 9 + j]):
                assert abs(rcov[h, j] - ocov[9 + h, 9 + j]) <= REL * scale


@pytest.mark.parametrize("V,d,how", [(200, 3, "classify"), (517, 2, "columns"), (64, 1, "classify")])
def test_cov_block_hard_call_fast_path(engine_factory, V, d, how, monkeypatch):
    """Hard-call blocks with an unweighted model take the exact int8 product (split over K when the band has few
    tiles): same numbers as the oracle and as the fp64 matrix-core path."""
    N = 2300
    G, chrom, pos, X, y = make_case(N, V, d, 0, 4000 + V)
    G = np.rint(G)
    G[:, 7] = 2.0                                            # monomorphic non-zero
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    eng = engine_factory()
    eng.set_null(0, X, res, np.full(N, s2), s2)
    eng.set_profiling(True)
    if how == "classify":
        ptr = eng.upload_block(G)
        assert eng.classify_block(ptr, V)
    else:                                                     # the adapter's way: per-column flags
        ptr = eng.alloc_block(V)
        for j in range(0, V, 50):
            eng.upload_columns(ptr, j, G[:, j:j + 50])
    cov, xz, zz, poly = eng.cov_block(ptr, V)
    monkeypatch.setenv("RVT_METACOV_FP64", "1")
    cov0, xz0, zz0, poly0 = eng.cov_block(ptr, V)
    monkeypatch.delenv("RVT_METACOV_FP64")
    rc, kept, ocov, row_end, oxz, ozz = orc.metacov(G, chrom, pos, X, y, 0, 10 ** 7)
    assert rc == 0 and (poly == kept).all() and (poly0 == kept).all()
    m = ~np.isnan(ocov)
    scale = np.abs(ocov[m]).max()
    assert np.abs(cov[m] - ocov[m]).max() < REL * scale
    assert np.abs(cov[m] - cov0[m]).max() < 1e-11 * scale
    kk = kept.astype(bool)
    assert np.allclose(xz[kk], oxz[kk], rtol=1e-9, atol=1e-9 * max(np.abs(oxz[kk]).max(), 1.0))
    assert np.allclose(zz, ozz, rtol=1e-9, atol=1e-9 * max(np.abs(ozz).max(), 1.0))


@pytest.mark.parametrize("binary", [0, 1])
def test_cov_block_fp64_gemm_against_the_one_wave_kernel(engine_factory, binary, monkeypatch):
    """Blocks that are not hard calls (dosages; any block under a binary trait) take the LDS-tiled fp64 product of round 5
    (gemm_f64.hip.h: 256 x 128 tiles, K split over the chip, N not a multiple of the 16-sample chunk, three row panels so
    that tiles below the diagonal are skipped and tiles across it are not).  Same band as round 4's path through the
    sufficient-statistics kernel (RVT_METACOV_PANEL=1) to rounding, and as the oracle."""
    N, V, d = 5003, 600, 3
    G, chrom, pos, X, y = make_case(N, V, d, binary, 9100 + binary)
    rng = np.random.default_rng(5)
    G = np.asfortranarray(np.clip(G + rng.uniform(-0.2, 0.2, size=G.shape) * (G > 0), 0.0, 2.0))   # dosages
    G[:, 11] = 0.5                                           # monomorphic
    if binary:
        rc, beta, p, v = orc.fit_logistic(X, y)
        res, s2 = y - p, 1.0
    else:
        rc, beta, pred, res, s2 = orc.fit_linear(X, y)
        v = np.full(N, s2)
    assert rc == 0
    eng = engine_factory()
    eng.set_null(binary, X, res, v, s2)
    ptr = eng.upload_block(G)
    cov, xz, zz, poly = eng.cov_block(ptr, V)
    monkeypatch.setenv("RVT_METACOV_PANEL", "1")
    cov0, xz0, zz0, poly0 = eng.cov_block(ptr, V)
    monkeypatch.delenv("RVT_METACOV_PANEL")
    monkeypatch.setenv("RVT_GEMM64_SLICES", "3")             # another split of the samples: same sums to rounding
    cov3 = eng.cov_block(ptr, V)[0]
    monkeypatch.delenv("RVT_GEMM64_SLICES")
    rc, kept, ocov, row_end, oxz, ozz = orc.metacov(G, chrom, pos, X, y, binary, 10 ** 7)
    assert rc == 0 and (poly == kept).all() and (poly0 == kept).all() and not poly[11]
    m = ~np.isnan(ocov)
    scale = np.abs(ocov[m]).max()
    assert np.abs(cov[m] - ocov[m]).max() < REL * scale
    assert np.abs(cov[m] - cov0[m]).max() < 1e-11 * scale
    assert np.abs(cov[m] - cov3[m]).max() < 1e-11 * scale
    kk = kept.astype(bool)
    assert np.allclose(xz[kk], oxz[kk], rtol=1e-9, atol=1e-9 * max(np.abs(oxz[kk]).max(), 1.0))
    assert np.allclose(xz[kk], xz0[kk], rtol=1e-11, atol=1e-11 * max(np.abs(xz0[kk]).max(), 1.0))


def _ring_fill(eng, ring, cap, col0, G, step=97):
    """Put the columns of G into a block used as a ring of `cap` columns, logical column j at physical (col0 + j) mod cap, the
    way the adapter does (rvt_block_upload_columns, a few columns at a time, split at the wrap)."""
    V = G.shape[1]
    j = 0
    while j < V:
        p = (col0 + j) % cap
        n = min(step, V - j, cap - p)
        eng.upload_columns(ring, p, G[:, j:j + n])
        j += n


def test_cov_band_wrapped_ring_against_the_oracle(engine_factory, monkeypatch):
    """The sliding window on a CIRCULAR ring (rvt_cov_band; RingMemoryPool, base/RingMemoryPool.cpp:31-63): 1 100 hard-call
    columns uploaded column-wise into a ring of 1 300 from physical column 900 on (the window wraps), N = 9 100 (several row
    slices of the column pass, K split over the chip), heads = every column, halo = 300 markers.  Against the ORACLE's rows
    (window = 300 positions apart), bit-identical to the rectangle call on the same columns laid out linearly, the same with
    the column cache off (the in-call int8 copy reads the wrapped columns), and with the float / (1/N) formatting applied on
    the device."""
    N, V, d, halo, cap, col0 = 9100, 1100, 3, 300, 1300, 900
    G, chrom, pos, X, y = make_case(N, V, d, 0, 4242)
    G = np.asfortranarray(np.rint(G))
    G[:, 17] = 0.0                                            # monomorphic
    G[:, 1050] = 2.0
    chrom = np.ones(V, dtype=np.int32)
    pos = np.arange(V, dtype=np.int32)
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    eng = engine_factory()
    eng.set_null(0, X, res, np.full(N, s2), s2)
    ring = eng.alloc_block(cap)
    _ring_fill(eng, ring, cap, col0, G)
    band, xz, zz, poly = eng.cov_band(ring, cap, col0, V, V, halo)
    assert eng.cov_band_last_path() == 1                      # MXFP4 band on the column cache
    rc, kept, ocov, row_end, oxz, ozz = orc.metacov(G, chrom, pos, X, y, 0, halo)
    assert rc == 0 and (poly == kept).all() and not poly[17] and not poly[1050]
    scale = np.nanmax(np.abs(ocov))
    checked = 0
    for h in range(V):
        if not kept[h]:
            continue
        js = np.arange(h, min(V, h + halo + 1))
        js = js[kept[js].astype(bool)]
        assert not np.isnan(ocov[h, js]).any() and row_end[h] == js[-1]
        assert np.abs(band[h, js - h].astype(np.float64) - ocov[h, js]).max() <= 2e-7 * scale   # (float32 band)
        checked += len(js)
        if h + halo + 1 > V:
            assert np.isnan(band[h, V - h:]).all()            # beyond the window: NaN
    assert checked > 250000
    kk = kept.astype(bool)
    assert np.allclose(xz[kk], oxz[kk], rtol=1e-9, atol=1e-9 * max(np.abs(oxz[kk]).max(), 1.0))
    assert np.allclose(zz, ozz, rtol=1e-9, atol=1e-9 * max(np.abs(ozz).max(), 1.0))
    # the same columns in a linear block through the rectangle call: the int8 products are exact either way -> bit-identical
    whole = eng.upload_block(G)
    rcov, rxz, rzz, rpoly = eng.cov_rect(whole, 0, V, V)
    for h in range(0, V, 13):
        w = min(halo + 1, V - h)
        assert np.array_equal(band[h, :w], rcov[h, h:h + w].astype(np.float32))
    assert np.array_equal(xz, rxz) and np.array_equal(poly, rpoly)
    # heads in the middle of the ring, a window that does not wrap, a window of one column
    for c0, H, W in ((col0 + 40, 500, 800), (5, 300, 400), (1299, 1, 1)):
        lo = (c0 - col0) % cap
        if lo + W > V:
            continue
        b2 = eng.cov_band(ring, cap, c0, H, W, halo)[0]
        for h in range(0, H, 7):
            w = min(halo + 1, W - h)
            assert np.array_equal(b2[h, :w], band[lo + h, :w])
    # the column cache off: the call makes its own int8 copy from the wrapped fp64 columns
    monkeypatch.setenv("RVT_METACOV_NO_CACHE", "1")
    band_nc = eng.cov_band(ring, cap, col0, V, V, halo)[0]
    monkeypatch.delenv("RVT_METACOV_NO_CACHE")
    assert np.array_equal(band_nc, band, equal_nan=True)
    # one byte per genotype on the int8 matrix instruction instead of E2M1 codes on the MXFP4 one: the same integers
    monkeypatch.setenv("RVT_BAND_INT8", "1")
    band_i8 = eng.cov_band(ring, cap, col0, V, V, halo)[0]
    assert eng.cov_band_last_path() == 11
    monkeypatch.setenv("RVT_METACOV_NO_CACHE", "1")
    band_i8_nc = eng.cov_band(ring, cap, col0, V, V, halo)[0]
    assert eng.cov_band_last_path() == 12
    monkeypatch.delenv("RVT_METACOV_NO_CACHE")
    monkeypatch.delenv("RVT_BAND_INT8")
    assert np.array_equal(band_i8, band, equal_nan=True) and np.array_equal(band_i8_nc, band, equal_nan=True)
    # K split differently: integers, the same sums
    monkeypatch.setenv("RVT_BAND_SLICES", "3")
    band_s3 = eng.cov_band(ring, cap, col0, V, V, halo)[0]
    monkeypatch.delenv("RVT_BAND_SLICES")
    assert np.array_equal(band_s3, band, equal_nan=True)
    # what the adapter prints: (float)value * (float)(1 / N), applied on the device
    sc = np.float32(1.0 / N)
    band_sc = eng.cov_band(ring, cap, col0, V, V, halo, scale=sc)[0]
    assert np.array_equal(band_sc, band * sc, equal_nan=True)


@pytest.mark.parametrize("binary", [0, 1])
def test_cov_band_wrapped_ring_fp64(engine_factory, binary):
    """Dosages (quantitative trait) and a binary trait's weights on the same circular ring: the band tiles on the fp64 matrix
    cores with the columns addressed modulo the capacity; 1 100 heads = two passes of 1 024.  Against the oracle."""
    N, V, d, halo, cap, col0 = 5003, 1100, 2, 150, 1200, 1000
    G, chrom, pos, X, y = make_case(N, V, d, binary, 777 + binary)
    if not binary:
        rng = np.random.default_rng(5)
        G = np.asfortranarray(np.clip(G + rng.uniform(-0.2, 0.2, size=G.shape) * (G > 0), 0.0, 2.0))   # dosages
    else:
        G = np.asfortranarray(np.rint(G))
    G[:, 30] = 1.0
    chrom = np.ones(V, dtype=np.int32)
    pos = np.arange(V, dtype=np.int32)
    if binary:
        rc, beta, p, v = orc.fit_logistic(X, y)
        res, s2 = y - p, 1.0
    else:
        rc, beta, pred, res, s2 = orc.fit_linear(X, y)
        v = np.full(N, s2)
    assert rc == 0
    eng = engine_factory()
    eng.set_null(binary, X, res, v, s2)
    ring = eng.alloc_block(cap)
    _ring_fill(eng, ring, cap, col0, G)
    band, xz, zz, poly = eng.cov_band(ring, cap, col0, V, V, halo)
    rc, kept, ocov, row_end, oxz, ozz = orc.metacov(G, chrom, pos, X, y, binary, halo)
    assert rc == 0 and (poly == kept).all() and not poly[30], (np.nonzero(poly != kept)[0], poly[poly != kept])
    scale = np.nanmax(np.abs(ocov))
    for h in range(V):
        if not kept[h]:
            continue
        js = np.arange(h, min(V, h + halo + 1))
        js = js[kept[js].astype(bool)]
        assert np.abs(band[h, js - h].astype(np.float64) - ocov[h, js]).max() <= 2e-7 * scale
    kk = kept.astype(bool)
    assert np.allclose(xz[kk], oxz[kk], rtol=1e-9, atol=1e-9 * max(np.abs(oxz[kk]).max(), 1.0))
    # against the rectangle call on a linear copy (another split of the samples: equal to rounding)
    whole = eng.upload_block(G)
    rcov = eng.cov_rect(whole, 0, 1024, V)[0]
    for h in range(0, 1024, 11):
        w = min(halo + 1, V - h)
        assert np.abs(band[h, :w] - rcov[h, h:h + w]).max() <= 2e-7 * scale


def test_cov_band_many_heads_two_passes(engine_factory):
    """More heads than one pass of the integer band takes (4 096): 4 500 hard-call columns, halo 40, a ring of 4 600 that
    wraps; rows against the oracle."""
    N, V, d, halo, cap, col0 = 4100, 4500, 1, 40, 4600, 4000
    rng = np.random.default_rng(99)
    G = np.asfortranarray(rng.binomial(2, rng.uniform(0.02, 0.4, V), size=(N, V)).astype(np.float64))
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=5)
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    chrom = np.ones(V, dtype=np.int32)
    pos = np.arange(V, dtype=np.int32)
    eng = engine_factory()
    eng.set_null(0, X, res, np.full(N, s2), s2)
    ring = eng.alloc_block(cap)
    _ring_fill(eng, ring, cap, col0, G, step=512)
    H = V - 10
    band, xz, zz, poly = eng.cov_band(ring, cap, col0, H, V, halo)
    rc, kept, ocov, row_end, oxz, ozz = orc.metacov(G, chrom, pos, X, y, 0, halo)
    assert rc == 0 and (poly == kept).all()
    scale = np.nanmax(np.abs(ocov))
    for h in range(H):
        js = np.arange(h, min(V, h + halo + 1))
        js = js[kept[js].astype(bool)]
        if kept[h]:
            assert np.abs(band[h, js - h].astype(np.float64) - ocov[h, js]).max() <= 2e-7 * scale


def test_stale_column_cache_is_not_used(engine_factory):
    """A column overwritten while hard calls are switched off (no column pass runs behind the upload) must not leave its OLD
    int8 copy / sums marked valid: after rvt_set_hardcall(1) a covariance call gives the numbers of the NEW data."""
    N, V, d = 3000, 130, 2
    G, chrom, pos, X, y = make_case(N, V, d, 0, 2024)
    G = np.asfortranarray(np.rint(G))
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    eng = engine_factory()
    eng.set_null(0, X, res, np.full(N, s2), s2)
    ring = eng.alloc_block(V)
    eng.upload_columns(ring, 0, G)
    first = eng.cov_block(ring, V)
    G2 = G.copy()
    G2[:, 40:60] = G[:, 60:80]                                # other hard calls in 20 columns
    eng.set_hardcall(False)
    eng.upload_columns(ring, 40, G2[:, 40:60])
    eng.set_hardcall(True)
    got = eng.cov_block(ring, V)
    ref = eng.cov_block(eng.upload_block(G2), V)
    iu = np.triu_indices(V)
    assert not np.array_equal(first[0][iu], ref[0][iu])
    scale = np.nanmax(np.abs(ref[0][iu]))
    assert np.abs(got[0][iu] - ref[0][iu]).max() <= 1e-11 * scale and np.array_equal(got[3], ref[3])
    # a copy inside one block invalidates what was known about its target columns as well
    eng.upload_columns(ring, 0, G)
    eng._check(eng.L.rvt_block_copy_columns(eng.ctx, C.c_void_p(ring), 40, C.c_void_p(ring), 60, 20))
    got = eng.cov_block(ring, V)
    assert np.abs(got[0][iu] - ref[0][iu]).max() <= 1e-11 * scale


def test_fp64_band_product_exact_integer_check():
    """tools/gemm64_bench check: C = A' D B of small-integer operands (dyadic weights: every sum exact whatever the order) on
    the fp64 matrix cores against a host product — symmetric / rectangular, weights, K not a multiple of the chunk, several K
    slices, the band enumeration (halo) and the ring addressing of the circular window."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gemm64_bench")
    assert os.path.exists(exe), "tools/gemm64_bench is built by __graft_entry__.build()"
    p = subprocess.run([exe, "check"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "all checks passed" in p.stdout, p.stdout + p.stderr
    assert p.stdout.count(": 0 /") == 9


def test_mxfp4_band_product_exact_integer_check():
    """tools/band_bench check: the band tiles of hard calls on v_mfma_scale_f32_32x32x64_f8f6f4 (E2M1 codes, unit block scales,
    fp32 accumulation of integers below 2^24) against the int8 kernel — every entry of the band — and both against plain dot
    products of the columns; rings that wrap, K not a multiple of the chunk, one / several K slices."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "band_bench")
    assert os.path.exists(exe), "tools/band_bench is built by __graft_entry__.build()"
    p = subprocess.run([exe, "check"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "all checks passed" in p.stdout, p.stdout + p.stderr
    assert p.stdout.count("fp4 != int8 in 0 /") == 5 and p.stdout.count("int8 0, fp4 0 of 512 wrong") == 5


def test_cov_band_mean_imputed_columns_stay_on_the_integer_band(engine_factory, monkeypatch):
    """What consolidate() leaves for real hard-call data: 0 / 1 / 2 plus ONE other value per column, the mean it imputed for
    the missing calls.  Uploaded a site at a time (as MetaCovTest::fit does) such a column crosses PCIe as 2-bit codes, the
    engine knows its other value, and the band is FOUR exact integer products on the MXFP4 instruction (h'h, h'm, m'h, m'm)
    combined with the mu's in fp64 — against the oracle's rows, against the fp64 band of the same ring
    (RVT_METACOV_FP64=1) to rounding, mixed with columns that have no missing call, on a ring that wraps, in two passes."""
    N, V, d, halo, cap, col0 = 6100, 1100, 2, 180, 1200, 1150
    G, chrom, pos, X, y = make_case(N, V, d, 0, 8080)
    G = np.asfortranarray(np.rint(G))
    rng = np.random.default_rng(17)
    n_imputed = 0
    for j in range(V):
        if j % 3 == 0:
            continue                                          # a third of the columns: no missing call
        miss = rng.random(N) < rng.choice([0.002, 0.02, 0.2])
        if miss.any() and not miss.all():
            G[miss, j] = G[~miss, j].mean()                   # imputeGenotypeToMean (DataConsolidator.cpp:217-245)
            n_imputed += 1
    G[:, 40] = 1.0                                            # monomorphic
    miss = rng.random(N) < 0.5
    G[miss, 41] = 0.0
    G[~miss, 41] = G[miss, 41].mean() + 0.25                  # a column that is one hard call and one other value only
    assert n_imputed > 600
    chrom = np.ones(V, dtype=np.int32)
    pos = np.arange(V, dtype=np.int32)
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    eng = engine_factory()
    eng.set_null(0, X, res, np.full(N, s2), s2)
    ring = eng.alloc_block(cap)
    for j in range(V):                                        # one site per call: the queued, packed upload
        eng.upload_columns(ring, (col0 + j) % cap, G[:, j])
    eng.set_profiling(True)
    band, xz, zz, poly = eng.cov_band(ring, cap, col0, V, V, halo)
    assert eng.cov_band_last_path() == 4                      # the four-product integer band, not the fp64 one
    rc, kept, ocov, row_end, oxz, ozz = orc.metacov(G, chrom, pos, X, y, 0, halo)
    assert rc == 0 and (poly == kept).all() and not poly[40]
    scale = np.nanmax(np.abs(ocov))
    worst = 0.0
    for h in range(V):
        if not kept[h]:
            continue
        js = np.arange(h, min(V, h + halo + 1))
        js = js[kept[js].astype(bool)]
        worst = max(worst, np.abs(band[h, js - h].astype(np.float64) - ocov[h, js]).max())
    assert worst <= 2e-7 * scale                              # (float32 band)
    kk = kept.astype(bool)
    assert np.allclose(xz[kk], oxz[kk], rtol=1e-9, atol=1e-9 * max(np.abs(oxz[kk]).max(), 1.0))
    # the same ring through the fp64 band: the integer decomposition agrees to rounding — and was actually taken above
    monkeypatch.setenv("RVT_METACOV_FP64", "1")
    band64 = eng.cov_band(ring, cap, col0, V, V, halo)[0]
    assert eng.cov_band_last_path() == 0
    monkeypatch.delenv("RVT_METACOV_FP64")
    m = ~np.isnan(band64)
    assert np.array_equal(np.isnan(band), np.isnan(band64)) and np.abs(band[m] - band64[m]).max() <= 2e-7 * scale
    # a window of columns without any other value inside the same ring takes the single product (bit-identical to a ring of
    # those columns alone) — and one dosage column sends its window to the fp64 band
    G2 = G.copy()
    G2[5, 300] = 0.3
    G2[6, 300] = 0.7                                          # two other values: not representable, crosses as doubles
    eng.upload_columns(ring, (col0 + 300) % cap, G2[:, 300])
    pure_cols = [(col0 + j) % cap for j in range(V) if j % 3 == 0]
    b3 = eng.cov_band(ring, cap, (col0 + 200) % cap, 150, 300, halo)[0]
    assert eng.cov_band_last_path() == 0                      # a window with a dosage column: the fp64 band
    eng.cov_band(ring, cap, (col0 + 400) % cap, 100, 250, halo)
    assert eng.cov_band_last_path() == 4                      # a window behind it: the integer band again
    rc, kept2, ocov2, _, _, _ = orc.metacov(G2, chrom, pos, X, y, 0, halo)
    for h in range(150):
        if kept2[200 + h]:
            js = np.arange(h, min(300, h + halo + 1))
            js = js[kept2[200 + js].astype(bool)]
            assert np.abs(b3[h, js - h].astype(np.float64) - ocov2[200 + h, 200 + js]).max() <= 2e-7 * scale


def _device_read(ptr, nbytes):
    """Read device memory behind the engine's back (hipMemcpy through the HIP runtime)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMemcpy.restype = C.c_int
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = np.empty(nbytes, dtype=np.uint8)
    assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(int(ptr)), nbytes, 2) == 0
    return out


@pytest.mark.parametrize("ncols", [64, 5, 1])
def test_uploaded_columns_are_what_the_block_holds(engine_factory, ncols):
    """rvt_block_upload_columns packs the caller's columns to 2-bit codes, sends them on one stream and expands them to doubles
    on another: uploads issued back to back — alternating between two matrices, into neighbouring column ranges, several columns
    at once (the immediate path) or one at a time (the queue) — must each land whole.  The block is read back with hipMemcpy."""
    N, d = 300_000, 2
    rng = np.random.default_rng(91)
    X = np.column_stack([np.ones(N), rng.normal(size=(N, d - 1))])
    y = rng.normal(size=N)
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    eng = engine_factory()
    eng.set_null(0, X, res, np.full(N, s2), s2)
    ld = eng.padded_ld(N)
    mats = []
    for k in range(2):
        A = (rng.random((N, ncols)) < 0.2).astype(np.float64) + (rng.random((N, ncols)) < 0.2)
        A[rng.random((N, ncols)) < 0.01] = 0.37 + k            # one other value per column (what an imputed mean is)
        mats.append(np.asfortranarray(A))
    reps = 10 if ncols > 1 else 70
    blk = eng.alloc_block(ncols * reps)
    for r in range(reps):
        eng.upload_columns(blk, ncols * r, mats[r % 2])
    eng.sync()
    got = _device_read(blk, 8 * ld * ncols * reps).view(np.float64).reshape(ncols * reps, ld)
    for r in range(reps):
        assert np.array_equal(got[ncols * r:ncols * (r + 1), :N], mats[r % 2].T), r
        assert not got[ncols * r:ncols * (r + 1), N:].any()


@pytest.mark.parametrize("imputed", [False, True])
def test_cov_band_at_the_bench_size_n500k(engine_factory, monkeypatch, imputed):
    """The window's integer band at BASELINE's sample count (N = 500 000: a K split over the whole chip, ~5 000 samples short of
    the 2^22-sample bound of an exact fp32 slice times none — every slice sum far above what a float mantissa holds in an
    inexact product): 300 hard-call columns (imputed: two thirds of them with 1 % of the calls replaced by the column mean)
    uploaded a site at a time into a ring that wraps.  Three checks that do not need the oracle on all of it: the integer
    band against the fp64 band of the same ring; 24 neighbouring columns against the ORACLE's rows on those columns alone (an
    entry depends on its two columns and the null model only); and the heads' order / flags."""
    N, V, d, halo, cap, col0 = 500_000, 300, 3, 120, 320, 250
    rng = np.random.default_rng(2026)
    X = np.column_stack([np.ones(N)] + [rng.normal(size=N) for _ in range(d - 1)])
    maf = rng.uniform(0.002, 0.3, V)
    G = np.empty((N, V), order="F")
    for j in range(V):
        G[:, j] = (rng.random(N) < maf[j]).astype(np.float64) + (rng.random(N) < maf[j])
    y = X @ rng.normal(size=d) + 0.05 * G[:, 7] + rng.normal(size=N)
    if imputed:
        for j in range(V):
            if j % 3:
                miss = rng.random(N) < 0.01
                G[miss, j] = G[~miss, j].mean()
    G[:, 11] = 2.0                                            # monomorphic
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    assert rc == 0
    eng = engine_factory()
    eng.set_null(0, X, res, np.full(N, s2), s2)
    ring = eng.alloc_block(cap)
    for j in range(V):
        eng.upload_columns(ring, (col0 + j) % cap, G[:, j])
    scale_f = np.float32(1.0 / N)
    band, xz, zz, poly = eng.cov_band(ring, cap, col0, V, V, halo, scale=scale_f)
    assert eng.cov_band_last_path() == (4 if imputed else 1)
    assert not poly[11] and poly.sum() == V - 1
    monkeypatch.setenv("RVT_METACOV_FP64", "1")
    band64 = eng.cov_band(ring, cap, col0, V, V, halo, scale=scale_f)[0]
    assert eng.cov_band_last_path() == 0
    monkeypatch.delenv("RVT_METACOV_FP64")
    m = ~np.isnan(band64)
    assert np.array_equal(np.isnan(band), np.isnan(band64))
    top = np.abs(band64[m]).max()
    assert np.abs(band[m] - band64[m]).max() <= 2e-7 * top    # (float32 rows of two fp64 computations)
    sub = np.arange(90, 114)
    chrom = np.ones(len(sub), dtype=np.int32)
    pos = np.arange(len(sub), dtype=np.int32)
    rc, kept, ocov, row_end, oxz, ozz = orc.metacov(np.asfortranarray(G[:, sub]), chrom, pos, X, y, 0, len(sub))
    assert rc == 0 and kept.all()
    for a, h in enumerate(sub):
        want = ocov[a, a:] * float(scale_f)
        got = band[h, :len(sub) - a].astype(np.float64)
        assert np.abs(got - want).max() <= 3e-7 * np.nanmax(np.abs(ocov)) * float(scale_f)
    assert np.allclose(xz[sub], oxz, rtol=1e-9, atol=1e-9 * max(np.abs(oxz).max(), 1.0))


def test_cov_band_does_not_depend_on_how_the_columns_were_uploaded(engine_factory):
    """Mean-imputed columns uploaded 97 at a time (packed, expanded and passed over in chunks of 32) leave the same cache as the
    same columns uploaded a site at a time: both rings take the four-product integer band and give the same bits."""
    N, V, d, halo, cap, col0 = 7001, 500, 2, 90, 640, 600
    G, chrom, pos, X, y = make_case(N, V, d, 0, 515)
    G = np.asfortranarray(np.rint(G))
    rng = np.random.default_rng(3)
    for j in range(0, V, 2):
        miss = rng.random(N) < 0.03
        G[miss, j] = G[~miss, j].mean()
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    eng = engine_factory()
    eng.set_null(0, X, res, np.full(N, s2), s2)
    r1, r2 = eng.alloc_block(cap), eng.alloc_block(cap)
    for j in range(V):
        eng.upload_columns(r1, (col0 + j) % cap, G[:, j])
    _ring_fill(eng, r2, cap, col0, G)
    b1, xz1, zz1, p1 = eng.cov_band(r1, cap, col0, V, V, halo)
    assert eng.cov_band_last_path() == 4
    b2, xz2, zz2, p2 = eng.cov_band(r2, cap, col0, V, V, halo)
    assert eng.cov_band_last_path() == 4
    assert np.array_equal(b1, b2, equal_nan=True) and np.array_equal(xz1, xz2) and np.array_equal(p1, p2)


def test_a_slot_rewritten_without_packing_leaves_no_mask_behind(engine_factory, monkeypatch):
    """A ring slot that held a mean-imputed column (cache state 2: hard calls + a mask of the other value) is overwritten by a
    column of pure hard calls that crosses as doubles (RVT_UPLOAD_FP64=1: the column pass of that path knows no other value).
    The window still mixes it with imputed columns, so the four-product band reads the slot's mask: it must be empty."""
    N, V, d, halo = 5200, 300, 2, 80
    G, chrom, pos, X, y = make_case(N, V, d, 0, 909)
    G = np.asfortranarray(np.rint(G))
    rng = np.random.default_rng(4)
    for j in range(V):
        miss = rng.random(N) < 0.05
        G[miss, j] = G[~miss, j].mean()
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    eng = engine_factory()
    eng.set_null(0, X, res, np.full(N, s2), s2)
    blk = eng.alloc_block(V)
    for j in range(V):
        eng.upload_columns(blk, j, G[:, j])
    G2 = G.copy()
    for j in (10, 11, 150):
        G2[:, j] = np.rint(rng.random(N) * 2.2).clip(0, 2)     # pure hard calls, no other value
    monkeypatch.setenv("RVT_UPLOAD_FP64", "1")
    for j in (10, 11, 150):
        eng.upload_columns(blk, j, G2[:, j])
    monkeypatch.delenv("RVT_UPLOAD_FP64")
    band = eng.cov_band(blk, 0, 0, V, V, halo)[0]
    assert eng.cov_band_last_path() == 4
    monkeypatch.setenv("RVT_METACOV_FP64", "1")
    band64 = eng.cov_band(blk, 0, 0, V, V, halo)[0]
    monkeypatch.delenv("RVT_METACOV_FP64")
    m = ~np.isnan(band64)
    assert np.array_equal(np.isnan(band), np.isnan(band64))
    assert np.abs(band[m] - band64[m]).max() <= 2e-7 * np.abs(band64[m]).max()
