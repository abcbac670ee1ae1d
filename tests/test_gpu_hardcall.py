"""GPU: the hard-call sufficient-statistics path (rvtests_amd/csrc/suffstat_hc.hip.h — int8 matrix cores for G'G, burden
collapse in the same pass) against the general fp64 path on the same blocks and against the oracle, including the
cases in which the in-pass burden collapse has to be redone (burden_fallback_kernel)."""
import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu

FIELDS = ("skat_Q", "skat_p", "skato_Q", "skato_p", "skato_rho", "cmc_U", "cmc_V", "cmc_stat", "cmc_p", "zeg_U",
          "zeg_V", "zeg_stat", "zeg_p")


def _hard_gene(N, M, seed, flip_col=None, ones_col=None, twos_col=None, zero_col=None, maf_hi=-1.0):
    rng = np.random.default_rng(seed)
    maf = 10 ** rng.uniform(-3.3, maf_hi, M)
    G = rng.binomial(2, maf, size=(N, M)).astype(np.float64)
    if flip_col is not None and M > flip_col:
        G[:, flip_col] = rng.binomial(2, 0.85, size=N)
    if ones_col is not None and M > ones_col:
        G[:, ones_col] = 1.0
    if twos_col is not None and M > twos_col:
        G[:, twos_col] = 2.0
    if zero_col is not None and M > zero_col:
        G[:, zero_col] = 0.0
    return np.asfortranarray(G), G.sum(0) / (2.0 * N)


def _same(f, x, y_, rel=1e-9):
    """int8 path against the fp64 path: `rel` relative; the 1-df burden statistics get an absolute floor as well — U is
    a sum that cancels to almost nothing for a null gene, and the weighted kernel's fixed-point null tile (42 bits per
    column, suffstat_hcx.hip.h) moves it by ~1e-12 of its natural scale, i.e. the chi-square statistic by ~1e-12 absolute"""
    floor = 1e-11 if f in ("cmc_stat", "zeg_stat", "cmc_U", "zeg_U") else 1e-300
    return abs(x - y_) <= rel * abs(y_) + floor


def _run(engine, genes, hard):
    """Run the genes starting on the hard-call kernel (hard=True: the default for blocks of unknown content) or with the
    engine confined to the general fp64 kernel (hard=False)."""
    ptrs = [engine.upload_block(G) for G, af in genes]
    engine.set_hardcall(hard)
    engine.set_profiling(True)
    engine.timing(reset=True)
    try:
        out = engine.run_blocks(ptrs, [G.shape[1] for G, af in genes], [af for G, af in genes])
        tm = engine.timing(reset=True)
    finally:
        engine.set_profiling(False)
        engine.set_hardcall(True)
    for p in ptrs:
        engine.free_block(p)
    return out, tm


@pytest.mark.parametrize("N,d", [(3000, 3), (4099, 1), (10007, 2)])
def test_hardcall_path_equals_general_path_and_oracle(engine, N, d):
    genes = [_hard_gene(N, M, seed=17 * M + d, flip_col=(2 if M % 3 == 0 else None),
                        twos_col=(5 if M % 4 == 1 else None), zero_col=(1 if M % 5 == 2 else None))
             for M in (1, 7, 16, 17, 30, 33, 48, 50, 64, 65, 80, 81, 96)]
    eff = 0.4 * genes[4][0][:, :3].sum(1)
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=5, G_effect=eff)
    engine.set_null(0, X, res, v, s2)
    hc, tm_hc = _run(engine, genes, True)
    gen, tm_gen = _run(engine, genes, False)
    assert tm_hc.genes == len(genes) and tm_hc.genes_hard_call == len(genes)       # the int8 path really ran
    assert tm_gen.genes_hard_call == 0
    for a, b, (G, af) in zip(hc, gen, genes):
        assert a.n_poly == b.n_poly and a.cmc_nonref == b.cmc_nonref and a.status == b.status
        for f in FIELDS:
            x, y_ = getattr(a, f), getattr(b, f)
            assert abs(x - y_) <= 1e-11 * abs(y_) + 1e-300, (f, x, y_)
        rc, o = orc.skat(G, af, X, res, v, 0)
        assert a.n_poly == o.n_poly
        if o.n_poly:
            assert abs(a.skat_Q - o.Q) <= 1e-10 * o.Q and abs(a.skat_p - o.pvalue) <= 1e-6 * o.pvalue + 1e-14
        rc3, c = orc.burden(G, X, y, 0, 0)
        if rc3 == 0:
            assert a.cmc_ok and a.cmc_nonref == c.nonref_site and abs(a.cmc_p - c.pvalue) <= 1e-6 * c.pvalue + 1e-14
        rc4, z = orc.burden(G, X, y, 0, 1)
        if rc4 == 0:
            assert a.zeg_ok and abs(a.zeg_p - z.pvalue) <= 1e-6 * z.pvalue + 1e-14


def test_burden_fallback_cases(engine):
    """The in-pass collapse assumes (a) a column is flipped iff its allele frequency > 1/2 and (b) no monomorphic
    column is counted.  An all-ones column breaks (b); an allele frequency that contradicts the column sum breaks (a).
    Both must come out exactly as the oracle has them (the engine redoes the burden sums of such a gene)."""
    N, d = 5003, 2
    g_ones = _hard_gene(N, 20, seed=1, ones_col=4)
    g_flip = _hard_gene(N, 35, seed=2, flip_col=7)
    g_ok = _hard_gene(N, 12, seed=3)
    wrong_af = g_flip[1].copy()
    wrong_af[7] = 0.01                 # says "rare" although the column sum exceeds N
    wrong_af[3] = 0.9                  # says "common" although the column is rare
    genes = [g_ones, (g_flip[0], wrong_af), g_ok]
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=8, G_effect=0.3 * g_ok[0][:, :4].sum(1))
    engine.set_null(0, X, res, v, s2)
    out, tm = _run(engine, genes, True)
    assert tm.genes_hard_call == 3
    for r, (G, af) in zip(out, genes):
        for which, ok, stat, p in ((0, r.cmc_ok, r.cmc_stat, r.cmc_p), (1, r.zeg_ok, r.zeg_stat, r.zeg_p)):
            rc, b = orc.burden(G, X, y, 0, which)
            assert ok == (rc == 0)
            if ok:
                assert abs(stat - b.stat) <= 1e-9 * b.stat + 1e-13 and abs(p - b.pvalue) <= 1e-6 * b.pvalue + 1e-14
        rc, c = orc.burden(G, X, y, 0, 0)
        assert r.cmc_nonref == c.nonref_site
        rc, a = orc.skat(G, af, X, res, v, 0)                   # the weights use the caller's af as given (quirk #3)
        assert r.n_poly == a.n_poly and abs(r.skat_Q - a.Q) <= 1e-10 * a.Q


def test_classification_query(engine):
    N = 2000
    X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=4)
    engine.set_null(0, X, res, v, s2)
    G, af = _hard_gene(N, 9, seed=6)
    p = engine.upload_block(G)
    assert engine.classify_block(p, 9) is True
    engine.free_block(p)
    for bad in (0.5, 3.0, -9.0, 1e-300, np.nan):
        H = G.copy()
        H[N - 1, 8] = bad
        p = engine.upload_block(H)
        assert engine.classify_block(p, 9) is False, bad
        engine.free_block(p)


def _exact_gram(G):
    """G'G with every product and sum exact (Fractions are overkill: the entries are doubles, use integer arithmetic on
    their 2^-k grid through Python's arbitrary precision)."""
    from fractions import Fraction
    N, M = G.shape
    S = np.empty((M, M))
    cols = [[Fraction(x) for x in G[:, j]] for j in range(M)]
    for a in range(M):
        for b in range(a, M):
            S[a, b] = S[b, a] = float(sum(x * y for x, y in zip(cols[a], cols[b])))
    return S


@pytest.mark.parametrize("N,d,miss", [(3000, 3, 0.02), (4099, 1, 0.3), (10007, 2, 0.001)])
def test_mean_imputed_columns_stay_on_the_hardcall_kernel(engine, N, d, miss):
    """imputeGenotypeToMean (src/DataConsolidator.cpp:217-245) leaves hard calls plus ONE other value per column.  Such
    genes run on the integer kernel (masked-entry tiles), are not handed back, and give the numbers of the general
    kernel and of the oracle; G'G equals the exactly rounded product."""
    genes = []
    for M in (1, 7, 16, 17, 30, 33, 48, 50, 64, 65, 80, 81, 96):
        Graw, Gi, afi = synth.make_gene(N, M, seed=3 * M + d, missing=miss, common=(M % 2 == 0), mono=(M % 3 == 0))
        genes.append((Gi, afi))
    assert any(((G != 0) & (G != 1) & (G != 2)).any() for G, af in genes)
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=5, G_effect=0.4 * genes[4][0][:, :3].sum(1))
    engine.set_null(0, X, res, v, s2)
    hc, tm_hc = _run(engine, genes, True)
    gen, tm_gen = _run(engine, genes, False)
    assert tm_hc.genes_hard_call == len(genes) and tm_hc.genes_handed_back == 0
    assert tm_gen.genes_hard_call == 0
    for a, b, (G, af) in zip(hc, gen, genes):
        assert a.n_poly == b.n_poly and a.cmc_nonref == b.cmc_nonref and a.status == b.status
        for f in FIELDS:
            # (the statistics agree to rounding; Davies' term count is a floor() of its inputs, so a p-value may move by
            # ~1e-7 relative under a last-bit change of G'G — DESIGN.md section 4)
            x, y_ = getattr(a, f), getattr(b, f)
            tol = 1e-6 if f.endswith("_p") else 1e-10
            assert abs(x - y_) <= tol * abs(y_) + 1e-300, (G.shape[1], f, x, y_)
        rc, o = orc.skat(G, af, X, res, v, 0)
        assert a.n_poly == o.n_poly
        if o.n_poly:
            assert abs(a.skat_Q - o.Q) <= 1e-10 * o.Q and abs(a.skat_p - o.pvalue) <= 1e-6 * o.pvalue + 1e-14
        rc2, so = orc.skato(G, af, X, res, v, 0)
        if rc2 == 0 and o.n_poly:
            assert abs(a.skato_p - so.pvalue) <= 1e-6 * so.pvalue + 5e-13
        for which, ok, p, nonref in ((0, a.cmc_ok, a.cmc_p, a.cmc_nonref), (1, a.zeg_ok, a.zeg_p, None)):
            rc3, c = orc.burden(G, X, y, 0, which)
            if rc3 == 0:
                assert ok and abs(p - c.pvalue) <= 1e-6 * c.pvalue + 1e-14
                if nonref is not None:
                    assert nonref == c.nonref_site
    # the Gram matrix itself: integer pieces exact, one rounding per product of the imputed value
    G, af = genes[3]
    ptr = engine.upload_block(G)
    S, T, u, colsum = engine.debug_suffstat(ptr, G.shape[1])[:4]
    engine.free_block(ptr)
    Sx = _exact_gram(G[:, :G.shape[1]])
    assert np.max(np.abs(S - Sx) / np.maximum(np.abs(Sx), 1.0)) <= 4 * np.finfo(float).eps
    assert np.allclose(T, G.T @ X, rtol=1e-12, atol=1e-9) and np.allclose(colsum, G.sum(0), rtol=1e-13)


def test_imputed_value_that_counts_in_the_burden_collapse(engine):
    """(int)g' > 0 for the imputed value itself: mu >= 1 in an unflipped column, mu <= 1 in a flipped one.  The in-pass
    collapse skips masked entries; such a gene must come out as the oracle has it (burden fallback)."""
    N, d = 5003, 2
    rng = np.random.default_rng(7)
    G, af = _hard_gene(N, 24, seed=5, maf_hi=-0.5)
    G = G.copy()
    col = rng.binomial(2, 0.52, N).astype(float)          # mean just above 1: unflipped when the sum stays <= N
    col[rng.random(N) < 0.05] = np.nan
    mu = np.nansum(col) / np.sum(~np.isnan(col))
    col[np.isnan(col)] = mu
    G[:, 3] = col
    col2 = rng.binomial(2, 0.9, N).astype(float)          # flipped column whose imputed value stays above 1
    col2[rng.random(N) < 0.03] = 1.8123
    G[:, 9] = col2
    G = np.asfortranarray(G)
    af = G.sum(0) / (2.0 * N)
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=8, G_effect=0.3 * G[:, :4].sum(1))
    engine.set_null(0, X, res, v, s2)
    (r,), tm = _run(engine, [(G, af)], True)
    assert tm.genes_hard_call == 1 and tm.genes_handed_back == 0
    for which, ok, stat, p in ((0, r.cmc_ok, r.cmc_stat, r.cmc_p), (1, r.zeg_ok, r.zeg_stat, r.zeg_p)):
        rc, b = orc.burden(G, X, y, 0, which)
        assert ok == (rc == 0)
        if ok:
            assert abs(stat - b.stat) <= 1e-9 * b.stat + 1e-13 and abs(p - b.pvalue) <= 1e-6 * b.pvalue + 1e-14
    rc, c = orc.burden(G, X, y, 0, 0)
    assert r.cmc_nonref == c.nonref_site
    rc, a = orc.skat(G, af, X, res, v, 0)
    assert r.n_poly == a.n_poly and abs(r.skat_Q - a.Q) <= 1e-10 * a.Q


def test_block_rewritten_in_place_with_dosages(engine):
    """Nothing is remembered about a block: the same device allocation first holds hard calls, then — rewritten in
    place, no call to the engine in between — dosages, then a column with two different non-integer values, then an
    infinity.  Every run must give the oracle's numbers (the integer kernel hands the gene back to the fp64 kernel)."""
    import ctypes as C
    N, M, d = 6007, 37, 2
    rng = np.random.default_rng(11)
    Gh, af = _hard_gene(N, M, seed=9, flip_col=4)
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=3, G_effect=0.3 * Gh[:, :3].sum(1))
    engine.set_null(0, X, res, v, s2)
    ptr = engine.upload_block(Gh)
    ld = int(engine.L.rvt_padded_ld(N))

    def rewrite(G):
        buf = np.zeros((ld, M), order="F")
        buf[:N] = G
        rc = _hip_memcpy_h2d(ptr, buf)
        assert rc == 0

    def check(G, handed_back):
        afx = G.sum(0) / (2.0 * N)
        engine.set_profiling(True)
        engine.timing(reset=True)
        (r,) = engine.run_blocks([ptr], [M], [afx])
        tm = engine.timing(reset=True)
        engine.set_profiling(False)
        assert tm.genes_hard_call == 1 and tm.genes_handed_back == handed_back
        rc, a = orc.skat(G, afx, X, res, v, 0)
        rc2, o = orc.skato(G, afx, X, res, v, 0)
        rc3, c = orc.burden(G, X, y, 0, 0)
        assert r.n_poly == a.n_poly and abs(r.skat_Q - a.Q) <= 1e-10 * a.Q
        assert abs(r.skat_p - a.pvalue) <= 1e-6 * a.pvalue + 1e-14 and abs(r.skato_p - o.pvalue) <= 1e-6 * o.pvalue + 5e-13
        assert r.cmc_nonref == c.nonref_site and abs(r.cmc_p - c.pvalue) <= 1e-6 * c.pvalue + 1e-14

    check(Gh, 0)
    Gd = np.clip(Gh + rng.normal(0, 0.05, Gh.shape), 0, 2)                 # dosages everywhere
    rewrite(Gd)
    check(Gd, 1)
    G2 = Gh.copy()                                                          # ONE column with two distinct fractions
    G2[17, 5] = 0.25
    G2[4000, 5] = 0.75
    rewrite(G2)
    check(G2, 1)
    G3 = Gh.copy()                                                          # one imputed value: stays
    G3[rng.random(N) < 0.01, 8] = 0.0371
    rewrite(G3)
    check(G3, 0)
    rewrite(Gh)
    check(Gh, 0)
    engine.free_block(ptr)


def _hip_memcpy_h2d(ptr, arr):
    """Write a host array into device memory behind the engine's back (hipMemcpy through the HIP runtime)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMemcpy.restype = C.c_int
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    a = np.ascontiguousarray(arr.T)           # column-major bytes
    return hip.hipMemcpy(C.c_void_p(int(ptr)), a.ctypes.data_as(C.c_void_p), a.nbytes, 1)


def test_content_hint_only_chooses_the_starting_kernel(engine):
    """Dosage blocks of the caller: without a hint they start on the integer kernel and are handed back gene by gene, with
    rvt_set_content_hint(0) they start on the fp64 kernel; hard calls under the wrong hint simply run on the fp64
    kernel.  The records are right every time, and equal between the two ways for dosages (same kernel computes them)."""
    N, M, d = 3001, 20, 2
    rng = np.random.default_rng(2)
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=13)
    engine.set_null(0, X, res, v, s2)
    dos = []
    for g in range(6):
        G = np.asfortranarray(np.clip(rng.binomial(2, 0.05, (N, M)) + rng.normal(0, 0.02, (N, M)), 0, 2))
        dos.append((G, G.sum(0) / (2.0 * N)))
    hard = [_hard_gene(N, M, seed=70 + g) for g in range(6)]
    pd = [engine.upload_block(G) for G, af in dos]
    ph = [engine.upload_block(G) for G, af in hard]
    engine.set_profiling(True)

    def run(ptrs, genes):
        engine.timing(reset=True)
        out = engine.run_blocks(ptrs, [M] * 6, [af for G, af in genes])
        tm = engine.timing(reset=True)
        for r, (G, af) in zip(out, genes):
            rc, a = orc.skat(G, af, X, res, v, 0)
            assert abs(r.skat_Q - a.Q) <= 1e-10 * a.Q and abs(r.skat_p - a.pvalue) <= 1e-6 * a.pvalue + 1e-14
        return out, (tm.genes_hard_call, tm.genes_handed_back)

    try:
        a, seen = run(pd, dos)
        assert seen == (6, 6)
        a2, seen = run(pd, dos)                       # no history: the same again
        assert seen == (6, 6)
        engine.set_content_hint(0)
        b, seen = run(pd, dos)
        assert seen == (0, 0)
        for x, y_ in zip(a, b):      # same fp64 statistics; the burden sums take another summation order
            assert all(getattr(x, f) == getattr(y_, f) for f in FIELDS if not f.startswith(("cmc", "zeg")))
            assert all(abs(getattr(x, f) - getattr(y_, f)) <= 1e-12 * abs(getattr(y_, f)) for f in FIELDS)
        h0, seen = run(ph, hard)                      # hard calls under the "dosages" hint: fp64 kernel, still right
        assert seen == (0, 0)
        engine.set_content_hint(-1)
        h1, seen = run(ph, hard)
        assert seen == (6, 0)
    finally:
        engine.set_content_hint(-1)
        engine.set_profiling(False)
    for p in pd + ph:
        engine.free_block(p)


def test_masked_tile_counters_at_the_wave_part_cap(engine, monkeypatch):
    """The masked tiles are 16-bit counters per wave-part; the host cuts wave-parts at 1020 steps (16 320 samples) so that
    P' = (H + 4m)'m <= 4 x 16 320 < 65 536.  A column that is imputed EVERYWHERE reaches that bound."""
    N = 70_000
    rng = np.random.default_rng(4)
    G = rng.binomial(2, 0.2, (N, 3)).astype(float)
    G[:, 1] = 0.4137                                  # every entry masked
    G[rng.random(N) < 0.5, 2] = 0.7311                # half the entries masked
    G = np.asfortranarray(G)
    af = G.sum(0) / (2.0 * N)
    X, y, res, v, s2 = synth.make_null(N, 1, 0, seed=6)
    engine.set_null(0, X, res, v, s2)
    monkeypatch.setenv("RVT_WPARTS", "1")             # as few wave-parts as the cap allows
    ptr = engine.upload_block(G)
    S = engine.debug_suffstat(ptr, 3)[0]
    engine.free_block(ptr)
    Sx = G.T.astype(np.longdouble) @ G.astype(np.longdouble)
    assert np.max(np.abs(S - Sx) / np.abs(Sx)) < 1e-14


@pytest.mark.parametrize("N,d", [(3000, 2), (4099, 1), (10007, 4)])
def test_binary_trait_weighted_hardcall_path(engine, N, d):
    """Binary trait: hard-call blocks of up to 80 variants run on the weighted int8 kernel (suffstat_hcw.hip.h: digit
    planes of v = p (1 - p)); wider ones stay on the fp64 kernel.  Both against the general path and the oracle."""
    Ms = (1, 7, 16, 17, 30, 33, 48, 50, 64, 65, 80, 81, 96)
    genes = [_hard_gene(N, M, seed=19 * M + d, flip_col=(2 if M % 3 == 0 else None),
                        twos_col=(5 if M % 4 == 1 else None), zero_col=(1 if M % 5 == 2 else None)) for M in Ms]
    eff = 0.5 * genes[4][0][:, :3].sum(1)
    X, y, res, v, s2 = synth.make_null(N, d, 1, seed=12, G_effect=eff)
    engine.set_null(1, X, res, v, s2)
    hc, tm_hc = _run(engine, genes, True)
    gen, tm_gen = _run(engine, genes, False)
    assert tm_hc.genes == len(genes) and tm_hc.genes_hard_call == sum(1 for M in Ms if M <= 80)
    assert tm_gen.genes_hard_call == 0
    for a, b, (G, af) in zip(hc, gen, genes):
        assert a.n_poly == b.n_poly and a.cmc_nonref == b.cmc_nonref and a.status == b.status
        for f in FIELDS:
            x, y_ = getattr(a, f), getattr(b, f)
            assert _same(f, x, y_), (G.shape[1], f, x, y_)
        rc, o = orc.skat(G, af, X, res, v, 1)
        assert a.n_poly == o.n_poly
        if o.n_poly:
            assert abs(a.skat_Q - o.Q) <= 1e-10 * o.Q and abs(a.skat_p - o.pvalue) <= 1e-6 * o.pvalue + 1e-14
        rc2, so = orc.skato(G, af, X, res, v, 1)
        if rc2 == 0 and o.n_poly:
            assert abs(a.skato_p - so.pvalue) <= 1e-6 * so.pvalue + 5e-13
        for which, ok, p in ((0, a.cmc_ok, a.cmc_p), (1, a.zeg_ok, a.zeg_p)):
            rc3, c = orc.burden(G, X, y, 1, which)
            if rc3 == 0:
                assert ok and abs(p - c.pvalue) <= 1e-6 * c.pvalue + 1e-14


def test_binary_trait_weights_at_the_digit_range_limits(engine):
    """v = 1/4 exactly (intercept-only model, balanced cases) is the largest weight the digit planes must hold; weights
    that are tiny or have all 49 fractional bits set exercise the balanced-digit carries."""
    N = 4000
    G, af = _hard_gene(N, 40, seed=31, flip_col=3)
    X = np.ones((N, 1), order="F")
    y = (np.arange(N) % 2).astype(np.float64)
    p0 = np.full(N, 0.5)
    res, v = y - p0, p0 * (1 - p0)
    engine.set_null(1, X, res, v, 1.0)
    (a,), tm = _run(engine, [(G, af)], True)
    (b,), _ = _run(engine, [(G, af)], False)
    assert tm.genes_hard_call == 1
    for f in FIELDS:
        assert _same(f, getattr(a, f), getattr(b, f)), f
    rng = np.random.default_rng(5)
    pr = np.concatenate([10.0 ** rng.uniform(-9, -1, N // 2), 1.0 - 10.0 ** rng.uniform(-9, -1, N - N // 2)])
    v2 = pr * (1 - pr)
    v2[:8] = [0.25, 0.25 - 2.0 ** -49, 2.0 ** -49, 2.0 ** -50, 2.0 ** -40, 127.0 / 512, 63.0 / 128 / 2, 1e-300]
    X2, y2, res2, _, s2 = synth.make_null(N, 2, 1, seed=9)
    engine.set_null(1, X2, res2, v2, 1.0)
    (a,), tm = _run(engine, [(G, af)], True)
    (b,), _ = _run(engine, [(G, af)], False)
    assert tm.genes_hard_call == 1
    for f in FIELDS:
        assert _same(f, getattr(a, f), getattr(b, f), rel=1e-8), f


def test_binary_trait_burden_fallback(engine):
    N, d = 5003, 2
    g_ones = _hard_gene(N, 20, seed=1, ones_col=4)
    g_flip = _hard_gene(N, 35, seed=2, flip_col=7)
    wrong_af = g_flip[1].copy()
    wrong_af[7] = 0.01
    wrong_af[3] = 0.9
    genes = [g_ones, (g_flip[0], wrong_af)]
    X, y, res, v, s2 = synth.make_null(N, d, 1, seed=8, G_effect=0.4 * g_ones[0][:, :4].sum(1))
    engine.set_null(1, X, res, v, s2)
    out, tm = _run(engine, genes, True)
    assert tm.genes_hard_call == 2
    for r, (G, af) in zip(out, genes):
        for which, ok, stat, p in ((0, r.cmc_ok, r.cmc_stat, r.cmc_p), (1, r.zeg_ok, r.zeg_stat, r.zeg_p)):
            rc, b = orc.burden(G, X, y, 1, which)
            assert ok == (rc == 0)
            if ok:
                assert abs(stat - b.stat) <= 1e-9 * b.stat + 1e-13 and abs(p - b.pvalue) <= 1e-6 * b.pvalue + 1e-14


def test_binary_trait_with_a_weight_out_of_range_stays_on_fp64(engine):
    N = 3000
    G, af = _hard_gene(N, 30, seed=11)
    X, y, res, v, s2 = synth.make_null(N, 2, 1, seed=12)
    v = v.copy()
    v[5] = 0.6                                  # not a logistic weight: no digit planes, general kernel
    engine.set_null(1, X, res, v, s2)
    out, tm = _run(engine, [(G, af)], True)
    assert tm.genes_hard_call == 0
    rc, o = orc.skato(G, af, X, res, v, 1)
    assert abs(out[0].skato_p - o.pvalue) <= 1e-6 * o.pvalue + 5e-13


def test_streaming_submissions_take_the_hardcall_path(engine):
    """Blocks written by rvt_submit_gene / _i8 / _bed start on the int8 path and give the oracle's numbers."""
    N, d = 4001, 3
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=21)
    engine.set_null(0, X, res, v, s2)
    genes = [_hard_gene(N, M, seed=40 + M, flip_col=(3 if M > 20 else None)) for M in (5, 18, 40, 70)]
    engine.set_profiling(True)
    engine.timing(reset=True)
    for i, (G, af) in enumerate(genes):
        engine.submit_gene(i, G, af)
    for i, (G, af) in enumerate(genes):
        engine.submit_gene_raw(10 + i, G.astype(np.int8), want_af=False)
    for i, (G, af) in enumerate(genes):
        engine.submit_gene_bed(20 + i, engine.pack_bed(G), G.shape[1], want_af=False)
    got = engine.collect()
    tm = engine.timing(reset=True)
    engine.set_profiling(False)
    assert [r.gene_id for r in got] == [0, 1, 2, 3, 10, 11, 12, 13, 20, 21, 22, 23]
    assert tm.genes_hard_call == 12
    for k, r in enumerate(got):
        G, af = genes[k % 4]
        rc, a = orc.skat(G, af, X, res, v, 0)
        rc2, o = orc.skato(G, af, X, res, v, 0)
        rc3, c = orc.burden(G, X, y, 0, 0)
        assert abs(r.skat_p - a.pvalue) <= 1e-6 * a.pvalue + 1e-14 and abs(r.skato_p - o.pvalue) <= 1e-6 * o.pvalue + 5e-13
        assert r.cmc_nonref == c.nonref_site and abs(r.cmc_p - c.pvalue) <= 1e-6 * c.pvalue + 1e-14


def test_collect_ready_returns_finished_prefix_without_draining(engine):
    """rvt_collect_ready: only finished genes, in submission order, nothing lost; rvt_collect gets the rest."""
    N, d = 3000, 2
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=77)
    engine.set_null(0, X, res, v, s2)
    genes = [_hard_gene(N, 5 + (g % 40), seed=900 + g) for g in range(70)]
    got = []
    for g, (G, af) in enumerate(genes):
        engine.submit_gene(500 + g, G, af)
        if g % 10 == 9:
            got += engine.collect_ready()
    assert len(got) <= 64                     # the last, incomplete group of 32 cannot have been launched
    got += engine.collect()
    assert [r.gene_id for r in got] == [500 + g for g in range(70)]
    for r, (G, af) in zip(got[::9], genes[::9]):
        rc, a = orc.skat(G, af, X, res, v, 0)
        assert abs(r.skat_p - a.pvalue) <= 1e-6 * a.pvalue + 1e-14


def test_weighted_hardcall_at_the_wave_part_cap(engine):
    """The weighted kernel keeps int32 pair sums across a whole wave-part; the host cuts wave-parts at 3072 steps
    (49 152 samples).  N = 6.4 million reaches that cap; weights whose digits are all near +63 and a column that is 2
    almost everywhere come close to the int32 range.  Against the fp64 kernel."""
    N = 6_400_000
    rng = np.random.default_rng(2)
    G = np.empty((N, 2), order="F")
    G[:, 0] = 2.0
    G[rng.integers(0, N, N // 100), 0] = 0.0
    G[:, 1] = rng.binomial(2, 0.3, N)
    af = G.sum(0) / (2.0 * N)
    X = np.ones((N, 1), order="F")
    vval = sum(d * 128.0 ** -(p + 1) for p, d in enumerate([62, 63, 63, 63, 63, 63]))
    assert 0.48 < vval < 0.49
    v = np.full(N, vval)
    res = rng.standard_normal(N) * 0.5
    res -= res.mean()
    engine.set_null(1, X, res, v, 1.0)
    (a,), tm = _run(engine, [(G, af)], True)
    (b,), _ = _run(engine, [(G, af)], False)
    assert tm.genes_hard_call == 1
    assert a.n_poly == b.n_poly == 2
    for f in FIELDS:
        x, y_ = getattr(a, f), getattr(b, f)
        assert _same(f, x, y_), (f, x, y_)


def test_weighted_hardcall_digit_rounding_at_small_p(engine):
    """The weighted kernel rounds every v = p (1 - p) to 42 fractional bits (six 7-bit digits).  Bound what that does to
    SMALL p-values at the size of BASELINE configs[3] (N = 200 000): causal genes whose SKAT / SKAT-O p-values lie in the
    decades 1e-7 .. 1e-13, the int8 kernel against the fp64 kernel on the same blocks (north_star: 1e-6 relative)."""
    N, d = 200_000, 2
    rng = np.random.default_rng(2026)
    X = np.asfortranarray(np.column_stack([np.ones(N), rng.standard_normal(N)]))
    genes = []
    eff = np.zeros(N)
    for k in range(10):
        M = (24, 50, 72)[k % 3]
        maf = 10 ** rng.uniform(-3.3, -1.3, M)
        G = np.asfortranarray(rng.binomial(2, maf, size=(N, M)).astype(np.float64))
        burden = G[:, :5].sum(1)
        ncp = 25.0 + 8.0 * k
        eff += 4.0 * np.sqrt(ncp / (burden.var() * N)) * (burden - burden.mean())
        genes.append((G, G.sum(0) / (2.0 * N)))
    pr = 1.0 / (1.0 + np.exp(-(-2.0 + 0.3 * X[:, 1] + eff)))
    y = (rng.random(N) < pr).astype(np.float64)
    engine.fit_null(1, X, y)
    hc, tm = _run(engine, genes, True)
    gen, _ = _run(engine, genes, False)
    assert tm.genes_hard_call == len(genes) and tm.genes_handed_back == 0
    small = 0
    worst = 0.0
    for a, b in zip(hc, gen):
        for f in ("skat_p", "skato_p"):
            x, y_ = getattr(a, f), getattr(b, f)
            if 1e-13 <= y_ <= 1e-7:
                small += 1
            if y_ >= 1e-13:
                worst = max(worst, abs(x - y_) / y_)
                assert abs(x - y_) <= 1e-6 * y_, (f, x, y_)
        assert abs(a.skato_Q - b.skato_Q) <= 1e-10 * abs(b.skato_Q)
    assert small >= 4, "the planted effects no longer reach the small decades: %d" % small
    print("weighted int8 vs fp64 kernel, p in [1e-13, 1]: max relative difference %.3g (%d values below 1e-7)" % (worst, small))


def test_cooperative_kernel_is_bit_reproducible_across_launch_shapes(monkeypatch):
    """The weighted cooperative kernel works in integers throughout: one launch for every class or one per class, the
    p-values on their own CUs or anywhere, 4 or 29 wave-parts per gene — the records are the same bit for bit (the per-gene
    stages reduce the wave-parts' integer partial sums in a fixed order)."""
    import rvtests_amd
    N, d = 20011, 3
    Ms = (3, 20, 33, 50, 64, 80)
    genes = [_hard_gene(N, M, seed=7 * M, flip_col=(2 if M > 40 else None)) for M in Ms]
    # two genes with mean-imputed entries (the sparse tables)
    for G, af in genes[2:4]:
        rng = np.random.default_rng(G.shape[1])
        for j in range(0, G.shape[1], 5):
            rows = rng.choice(N, 9, replace=False)
            keep = np.ones(N, dtype=bool)
            keep[rows] = False
            G[rows, j] = G[keep, j].mean()
    X, y, res, v, s2 = synth.make_null(N, d, 1, seed=3, G_effect=0.4 * genes[3][0][:, :3].sum(1))
    recs = {}
    for tag, env in (("default", {}), ("per_class", {"RVT_HCX_FUSED": "0"}), ("pv_anywhere", {"RVT_PV_CUS": "0"}),
                     ("parts29", {"RVT_WPARTS": "29"})):
        for k in ("RVT_HCX_FUSED", "RVT_PV_CUS", "RVT_WPARTS"):
            monkeypatch.delenv(k, raising=False)
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        eng = rvtests_amd.Engine(0)
        eng.set_null(1, X, res, v, s2)
        out, tm = _run(eng, genes, True)
        assert tm.genes_hard_call == len(genes) and tm.genes_handed_back == 0
        recs[tag] = out
        eng.close()
    for tag in ("per_class", "pv_anywhere"):
        for a, b in zip(recs["default"], recs[tag]):
            for f in FIELDS:
                assert getattr(a, f) == getattr(b, f), (tag, f)
    for a, b in zip(recs["default"], recs["parts29"]):       # (another split: the fp64 per-gene reductions see other partials)
        for f in FIELDS:
            assert _same(f, getattr(a, f), getattr(b, f), rel=1e-11), ("parts29", f)


def test_binary_trait_with_an_outlier_covariate_keeps_the_exact_null_tile(monkeypatch):
    """The workgroup-cooperative binary-trait kernel multiplies the null tile as 42-bit fixed point below each column's LARGEST
    entry.  A covariate with one entry seven orders of magnitude above the rest would leave its typical entries a few bits: such
    a model must stay on the one-wave kernel (fp64 products of the tile) — the records then equal those with RVT_HCX=0 bit for
    bit (round 6: the test against the root mean square let this model through, the outlier drags the rms along)."""
    import rvtests_amd
    N, d = 20000, 3
    rng = np.random.default_rng(44)
    X = np.column_stack([np.ones(N), rng.normal(size=N), rng.normal(size=N)])
    X[123, 2] = 1e7
    eta = -1.0 + 0.4 * X[:, 1]
    y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(np.float64)
    rc, beta, p, v = orc.fit_logistic(np.asfortranarray(X), y)
    assert rc == 0
    genes = [_hard_gene(N, M, seed=60 + M) for M in (20, 48, 70)]
    outs = []
    for hcx in (None, "0"):
        if hcx is None:
            monkeypatch.delenv("RVT_HCX", raising=False)
        else:
            monkeypatch.setenv("RVT_HCX", hcx)
        e = rvtests_amd.Engine(0)
        e.set_null(1, np.asfortranarray(X), y - p, v, 1.0)
        ptrs = [e.upload_block(G) for G, af in genes]
        out = e.run_blocks(ptrs, [G.shape[1] for G, af in genes], [af for G, af in genes])
        outs.append([(r.skat_Q, r.skat_p, r.skato_p, r.cmc_U, r.cmc_p, r.zeg_stat) for r in out])
        e.close()
    assert outs[0] == outs[1]
    rc, a = orc.skat(genes[1][0], genes[1][1], np.asfortranarray(X), y - p, v, 1)
    assert abs(outs[0][1][0] - a.Q) <= 1e-9 * a.Q
