"""GPU: genes kept as PLINK 2-bit rows on the device (rvtests_amd/csrc/suffstat_hcp.hip.h, rvt_submit_gene_bed) against the
same genes expanded to fp64 blocks (RVT_PACKED=0 / the block entry points) and against the oracle: missing calls in every
shape (none, sparse, a whole column, a column of one value plus missing), flipped and monomorphic columns, every tile class,
sample counts that are not multiples of 4 / 16 / 64, and tests the packed kernel does not serve (permutations)."""
import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu

FIELDS = ("skat_Q", "skat_p", "skato_Q", "skato_p", "skato_rho", "cmc_U", "cmc_V", "cmc_stat", "cmc_p", "zeg_U",
          "zeg_V", "zeg_stat", "zeg_p", "cmc_nonref", "n_poly", "status")


def _raw_gene(N, M, seed, missing=0.01):
    rng = np.random.default_rng(seed)
    maf = 10 ** rng.uniform(-2.7, -0.5, M)
    raw = rng.binomial(2, maf, size=(N, M)).astype(np.float64)
    if missing > 0:
        raw[rng.random((N, M)) < missing] = -9.0
    if M > 3:
        raw[:, 1] = np.where(raw[:, 1] < 0, -9.0, rng.binomial(2, 0.9, size=N))      # flipped column (sum > N)
    if M > 6:
        raw[:, 4] = np.where(rng.random(N) < 0.1, -9.0, 1.0)                        # one value + missing: monomorphic
    if M > 8:
        raw[:, 7] = 0.0                                                             # all zero
    if M > 10:
        raw[:, 9] = -9.0                                                            # nothing but missing calls
    if M > 12:
        raw[:, 11] = np.where(rng.random(N) < 0.3, -9.0, rng.binomial(2, 0.7, size=N))  # imputed value >= 1: it counts
    return np.asfortranarray(raw)


@pytest.mark.parametrize("N,d", [(3000, 2), (4099, 1), (10007, 3), (61, 1), (130, 2)])
def test_packed_rows_equal_the_expanded_blocks_and_the_oracle(N, d, monkeypatch):
    import rvtests_amd
    genes = [_raw_gene(N, M, seed=31 * M + d, missing=(0.0 if M % 4 == 0 else 0.01))
             for M in (1, 5, 16, 17, 30, 33, 48, 50, 64, 65, 80, 81, 96)]
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=5, G_effect=0.4 * orc.impute_mean(genes[4])[:, :3].sum(1))
    outs = {}
    for packed in ("1", "0"):
        monkeypatch.setenv("RVT_PACKED", packed)
        # (the switch is read once per process: a fresh library handle would not re-read it — use the engine's own knob)
        e = rvtests_amd.Engine(0)
        e.set_null(0, X, res, v, s2)
        if packed == "0":
            e.set_hardcall(True)
        e.set_profiling(True)
        e.timing(reset=True)
        afs = []
        for g, raw in enumerate(genes):
            if packed == "1":
                afs.append(e.submit_gene_bed(g, e.pack_bed(raw), raw.shape[1]))
            else:
                afs.append(e.submit_gene_raw(g, raw.astype(np.int8)))
        outs[packed] = (e.collect(), afs, e.timing(reset=True))
        e.close()
    got, af_p, tm = outs["1"]
    ref, af_r, _ = outs["0"]
    assert tm.genes_hard_call == len(genes) and tm.genes_handed_back == 0
    for a, b, fa, fb, raw in zip(got, ref, af_p, af_r, genes):
        assert np.array_equal(fa, fb)
        for f in FIELDS:
            x, y_ = getattr(a, f), getattr(b, f)
            if f.startswith("cmc") and f != "cmc_nonref" and abs(b.cmc_U) < 1e-8:
                continue      # every sample counts: the CMC genotype is the constant 1 and U = the sum of the residuals = rounding
            assert x == y_ or (x != x and y_ != y_) or abs(x - y_) <= 1e-11 * abs(y_), (raw.shape[1], f, x, y_)
        G = orc.impute_mean(raw)
        rc, o = orc.skat(G, orc.counter_af(raw), X, res, v, 0)
        assert a.n_poly == o.n_poly
        if o.n_poly:
            assert abs(a.skat_Q - o.Q) <= 1e-10 * o.Q and abs(a.skat_p - o.pvalue) <= 1e-6 * o.pvalue + 1e-14
        rc3, c = orc.burden(G, X, y, 0, 0)
        if rc3 == 0:
            assert a.cmc_nonref == c.nonref_site
            if abs(b.cmc_U) >= 1e-8:        # (not the constant genotype, whose variance is rounding noise on both sides)
                assert a.cmc_ok and abs(a.cmc_p - c.pvalue) <= 1e-6 * c.pvalue + 1e-14
        rc4, z = orc.burden(G, X, y, 0, 1)
        if rc4 == 0:
            assert a.zeg_ok and abs(a.zeg_stat - z.stat) <= 1e-9 * z.stat + 1e-13


def test_tests_that_need_the_block_expand_it(engine):
    """SKAT with permutations reads the fp64 block: such a gene is expanded as before; a binary trait likewise."""
    import rvtests_amd
    N = 2000
    raw = _raw_gene(N, 20, seed=3)
    X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=4)
    engine.set_null(0, X, res, v, s2)
    prm = rvtests_amd.Params.default()
    prm.skat_nperm = 200
    engine.submit_gene_bed(0, engine.pack_bed(raw), 20, params=prm, want_af=False)
    engine.submit_gene_bed(1, engine.pack_bed(raw), 20, want_af=False)
    a, b = engine.collect()
    assert a.skat_Q == b.skat_Q and a.n_poly == b.n_poly and a.skat_p == b.skat_p
    Xb, yb, resb, vb, s2b = synth.make_null(N, 2, 1, seed=4)
    engine.set_null(1, Xb, resb, vb, s2b)
    engine.submit_gene_bed(0, engine.pack_bed(raw), 20, want_af=False)
    engine.submit_gene_raw(1, raw.astype(np.int8), want_af=False)
    a, b = engine.collect()
    for f in FIELDS:
        assert getattr(a, f) == getattr(b, f), f


@pytest.mark.parametrize("N,binary", [(4099, 0), (10007, 0), (3001, 1)])
def test_genes_of_a_resident_bed_matrix_give_the_records_of_the_host_rows(engine, N, binary):
    """rvt_bed_alloc / rvt_bed_upload / rvt_submit_gene_bed_dev: a .bed matrix kept in device memory in the file's own layout
    (rows of ceil(N/4) bytes, no padding: rows start at odd addresses for these N), genes named by the device address of their
    first row — one at a time with allele frequencies, several per call, with permutations (the gene is expanded) and under a
    binary trait (expanded too).  Every record and frequency equals the one rvt_submit_gene_bed gives for the same rows."""
    import rvtests_amd
    Ms = (1, 5, 17, 30, 48, 64, 81, 96, 120)                 # (120: wider than the packed-row kernel takes — expanded)
    genes = [_raw_gene(N, M, seed=77 * M + 1, missing=(0.0 if M % 2 else 0.02)) for M in Ms]
    X, y, res, v, s2 = synth.make_null(N, 2, binary, seed=12)
    engine.set_null(binary, X, res, v, s2)
    rows = [engine.pack_bed(g) for g in genes]
    cb = (N + 3) // 4
    first = np.concatenate([[0], np.cumsum(Ms)[:-1]]) + 3     # (three rows of something else in front)
    total = int(first[-1] + Ms[-1] + 2)
    d_bed = engine.bed_alloc(total)
    engine.bed_upload(d_bed, 0, np.full((3, cb), 0x55, dtype=np.uint8))
    for f, r in zip(first, rows):
        engine.bed_upload(d_bed, int(f), r)
    prm = rvtests_amd.Params.default()
    prm.skat_nperm = 100
    want_af, want = [], []
    for g, (r, M) in enumerate(zip(rows, Ms)):
        want_af.append(engine.submit_gene_bed(g, r, M))
    engine.submit_gene_bed(100, rows[3], Ms[3], params=prm, want_af=False)
    want = engine.collect()
    got_af = [engine.submit_gene_bed_dev(g, d_bed + int(first[g]) * cb, M) for g, M in enumerate(Ms)]
    engine.submit_gene_bed_dev(100, d_bed + int(first[3]) * cb, Ms[3], params=prm, want_af=False)
    got = engine.collect()
    engine.submit_genes_bed_dev(list(range(len(Ms))), [d_bed + int(f) * cb for f in first], Ms)
    got_many = engine.collect()
    engine.bed_free(d_bed)
    assert len(got) == len(want) == len(Ms) + 1 and len(got_many) == len(Ms)
    for a, b in zip(got_af, want_af):
        assert np.array_equal(a, b)
    # (resident genes that stay packed form G'[X | rr] on the int8 matrix cores from digit planes of the null tile: 1e-13 from the
    #  fp64 product of the host hand-off — and SKAT-O's p, Davies' method inside a quadrature, moves at the 1e-7 level with the
    #  last bits of its inputs; expanded genes (permutations, a binary trait, M > 96) are bit-identical)
    widths = list(Ms) + [None] + list(Ms)                      # (None: the gene with permutations)
    for k, (a, b) in enumerate(list(zip(got, want)) + list(zip(got_many, want[:len(Ms)]))):
        exact = binary or widths[k] is None or widths[k] > 96
        for f in FIELDS:
            x, y_ = getattr(a, f), getattr(b, f)
            if exact or f in ("cmc_nonref", "n_poly", "status"):
                assert x == y_ or (x != x and y_ != y_), (k, f, x, y_)
            elif f.startswith("cmc") and abs(b.cmc_U) < 1e-8:
                continue
            else:
                assert x == y_ or (x != x and y_ != y_) or abs(x - y_) <= (1e-6 if f == "skato_p" else 1e-11) * abs(y_), (k, f, x, y_)
    if not binary:
        assert want[3].skat_p > 0


def test_int8_genes_are_packed_on_the_way(engine, monkeypatch):
    """rvt_submit_gene_i8 at a size where the staging threads pack (N >= 4096): the gene crosses PCIe as .bed rows and gives the
    records and allele frequencies of rvt_submit_gene_bed on the same calls bit for bit, and those of the unpacked int8
    hand-off (RVT_PACK_I8=0) to rounding; a gene with a value above 2 is not a hard-call gene and crosses as bytes."""
    N = 9001
    Ms = (5, 30, 64, 81)
    genes = [_raw_gene(N, M, seed=5 * M + 2, missing=0.02) for M in Ms]
    X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=21)
    engine.set_null(0, X, res, v, s2)
    engine.set_profiling(True)
    af_bed = [engine.submit_gene_bed(g, engine.pack_bed(raw), raw.shape[1]) for g, raw in enumerate(genes)]
    want = engine.collect()
    engine.timing(reset=True)
    af_i8 = [engine.submit_gene_raw(g, raw.astype(np.int8)) for g, raw in enumerate(genes)]
    got = engine.collect()
    tm = engine.timing(reset=True)
    assert tm.genes_hard_call == len(genes)
    monkeypatch.setenv("RVT_PACK_I8", "0")
    af_un = [engine.submit_gene_raw(g, raw.astype(np.int8)) for g, raw in enumerate(genes)]
    plain = engine.collect()
    monkeypatch.delenv("RVT_PACK_I8")
    for a, b, c_, fa, fb, fc in zip(got, want, plain, af_i8, af_bed, af_un):
        assert np.array_equal(fa, fb) and np.array_equal(fa, fc)
        for f in FIELDS:
            x, y_, z = getattr(a, f), getattr(b, f), getattr(c_, f)
            assert x == y_ or (x != x and y_ != y_), (f, x, y_)
            if f.startswith("cmc") and f != "cmc_nonref" and abs(c_.cmc_U) < 1e-8:
                continue
            assert x == z or abs(x - z) <= 1e-11 * abs(z), (f, x, z)
    odd = genes[1].astype(np.int8)
    odd[17, 3] = 3                                            # not a hard call: PLINK's codes cannot say it
    engine.submit_gene_raw(0, odd, want_af=False)
    monkeypatch.setenv("RVT_PACK_I8", "0")
    engine.submit_gene_raw(1, odd, want_af=False)
    monkeypatch.delenv("RVT_PACK_I8")
    a, b = engine.collect()
    for f in FIELDS:
        assert getattr(a, f) == getattr(b, f), f


def test_resident_bed_genes_at_the_bench_size(engine):
    """N = 500 000 (rows of 125 000 bytes; the batch path of rvt_submit_genes kind 7): two genes of a resident matrix against the
    same rows handed over from host memory and against the int8 hand-off packed on the way — identical records."""
    N = 500_000
    Ms = (50, 77)
    genes = [_raw_gene(N, M, seed=3 * M, missing=0.01) for M in Ms]
    X, y, res, v, s2 = synth.make_null(N, 3, 0, seed=33)
    engine.set_null(0, X, res, v, s2)
    rows = [engine.pack_bed(g) for g in genes]
    cb = (N + 3) // 4
    d_bed = engine.bed_alloc(sum(Ms))
    engine.bed_upload(d_bed, 0, rows[0])
    engine.bed_upload(d_bed, Ms[0], rows[1])
    for g, (r, M) in enumerate(zip(rows, Ms)):
        engine.submit_gene_bed(g, r, M, want_af=False)
    want = engine.collect()
    engine.submit_genes_bed_dev([0, 1], [d_bed, d_bed + Ms[0] * cb], Ms)
    got = engine.collect()
    for g, raw in enumerate(genes):
        engine.submit_gene_raw(g, raw.astype(np.int8), want_af=False)
    got8 = engine.collect()
    engine.bed_free(d_bed)
    # the resident records against the ORACLE at this size (gene 0: SKAT's Q and p, both burden tests)
    G0 = orc.impute_mean(genes[0])
    rc, o = orc.skat(G0, orc.counter_af(genes[0]), X, res, v, 0)
    assert rc == 0 and got[0].n_poly == o.n_poly
    assert abs(got[0].skat_Q - o.Q) <= 1e-10 * o.Q and abs(got[0].skat_p - o.pvalue) <= 1e-6 * o.pvalue + 1e-14
    for which, stat, pv in ((0, got[0].cmc_stat, got[0].cmc_p), (1, got[0].zeg_stat, got[0].zeg_p)):
        rcb, b_ = orc.burden(G0, X, y, 0, which)
        assert rcb == 0 and abs(stat - b_.stat) <= 1e-9 * b_.stat + 1e-13 and abs(pv - b_.pvalue) <= 1e-6 * b_.pvalue + 1e-14
    for a, b, c_ in zip(got, want, got8):
        assert a.skat_p > 0 and a.n_poly > 0
        for f in FIELDS:
            assert getattr(c_, f) == getattr(b, f), f              # (the two host hand-offs: the same kernels, the same bits)
            x, y_ = getattr(a, f), getattr(b, f)                   # (resident: G'[X | rr] from digit planes, see above)
            assert x == y_ or abs(x - y_) <= (1e-6 if f == "skato_p" else 1e-11) * abs(y_), (f, x, y_)


def test_resident_genes_with_a_badly_scaled_covariate(engine):
    """The digit planes of the null tile carry 56 bits below twice a column's LARGEST entry: a covariate with one entry five
    orders of magnitude above the rest leaves its typical entries ~40 bits — the resident records still agree with the host
    hand-off's (fp64 product) far inside the parity tolerance; a column whose largest entry exceeds 2^16 x the median of its
    non-zero magnitudes is refused and the fp64 product is used (bit-identical records then)."""
    N = 9001
    Ms = (30, 64)
    genes = [_raw_gene(N, M, seed=11 * M, missing=0.01) for M in Ms]
    rows = [engine.pack_bed(g) for g in genes]
    cb = (N + 3) // 4
    rng = np.random.default_rng(5)
    for outlier, exact in ((3e4, False), (1e9, True)):
        X = np.column_stack([np.ones(N), rng.normal(size=N), rng.normal(size=N)])
        X[17, 2] = outlier
        y = X[:, 1] * 0.3 + rng.normal(size=N)
        rc, beta, pred, res, s2 = orc.fit_linear(np.asfortranarray(X), y)
        assert rc == 0
        engine.set_null(0, np.asfortranarray(X), res, np.full(N, s2), s2)
        d_bed = engine.bed_alloc(sum(Ms))
        engine.bed_upload(d_bed, 0, rows[0])
        engine.bed_upload(d_bed, Ms[0], rows[1])
        for g, (r, M) in enumerate(zip(rows, Ms)):
            engine.submit_gene_bed(g, r, M, want_af=False)
        want = engine.collect()
        engine.submit_genes_bed_dev([0, 1], [d_bed, d_bed + Ms[0] * cb], Ms)
        got = engine.collect()
        engine.bed_free(d_bed)
        for a, b in zip(got, want):
            assert b.skat_p > 0
            for f in FIELDS:
                x, y_ = getattr(a, f), getattr(b, f)
                if exact:
                    assert x == y_, (outlier, f, x, y_)
                else:
                    assert x == y_ or abs(x - y_) <= (1e-6 if f == "skato_p" else 1e-9) * abs(y_), (outlier, f, x, y_)


def test_resident_bed_entry_points_refuse_what_they_cannot_take():
    """Error behaviour of the resident .bed calls: no null model yet (it defines the row length), bad arguments, an empty list."""
    import rvtests_amd
    e = rvtests_amd.Engine(0)
    with pytest.raises(rvtests_amd.RvtError):
        e.bed_alloc(10)                                       # no null model: N unknown
    N = 5000
    X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=1)
    e.set_null(0, X, res, v, s2)
    with pytest.raises(rvtests_amd.RvtError):
        e.bed_alloc(0)
    d_bed = e.bed_alloc(40)
    raw = _raw_gene(N, 40, seed=9)
    e.bed_upload(d_bed, 0, e.pack_bed(raw))
    with pytest.raises(rvtests_amd.RvtError):
        e.submit_gene_bed_dev(0, d_bed, 0)                    # M = 0
    with pytest.raises(rvtests_amd.RvtError):
        e.submit_gene_bed_dev(0, 0, 5)                        # null address
    e.submit_genes_bed_dev([], [], [])                        # nothing: fine
    assert e.collect() == []
    af = e.submit_gene_bed_dev(7, d_bed, 40)
    (r,) = e.collect()
    assert r.gene_id == 7 and np.array_equal(af, orc.counter_af(raw))
    e.bed_free(d_bed)
    e.bed_free(0)                                             # a null pointer: nothing to free
    e.close()


@pytest.mark.parametrize("N", [8000, 8001, 16016])
def test_resident_rows_of_whole_dwords_take_the_same_records(engine, N):
    """ceil(N/4) a multiple of 4 (N = 8 000, 16 016): the batched count-and-copy pass moves 16 samples per thread; N = 8 001: bytes.
    Records and counts of resident genes equal the host hand-off's either way (expanded genes: bit for bit)."""
    import rvtests_amd
    Ms = (7, 33, 64)
    genes = [_raw_gene(N, M, seed=3 * M + N % 7, missing=0.02) for M in Ms]
    X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=2)
    engine.set_null(0, X, res, v, s2)
    rows = [engine.pack_bed(g) for g in genes]
    cb = (N + 3) // 4
    first = np.concatenate([[0], np.cumsum(Ms)[:-1]])
    d_bed = engine.bed_alloc(int(sum(Ms)))
    for f, r in zip(first, rows):
        engine.bed_upload(d_bed, int(f), r)
    prm = rvtests_amd.Params.default()
    prm.skat_nperm = 50                                       # expanded genes: the same kernels on the same doubles
    for g, (r, M) in enumerate(zip(rows, Ms)):
        engine.submit_gene_bed(g, r, M, params=prm, want_af=False)
    want = engine.collect()
    engine.submit_genes_bed_dev(list(range(len(Ms))), [d_bed + int(f) * cb for f in first], Ms, params=prm)
    got = engine.collect()
    for a, b in zip(got, want):
        for f in FIELDS:
            assert getattr(a, f) == getattr(b, f), f
    engine.submit_genes_bed_dev(list(range(len(Ms))), [d_bed + int(f) * cb for f in first], Ms)
    got2 = engine.collect()
    for g, raw in enumerate(genes):
        engine.submit_gene_bed(g, rows[g], Ms[g], want_af=False)
    want2 = engine.collect()
    for a, b in zip(got2, want2):
        assert a.n_poly == b.n_poly and a.cmc_nonref == b.cmc_nonref
        assert abs(a.skat_Q - b.skat_Q) <= 1e-11 * abs(b.skat_Q) and abs(a.skat_p - b.skat_p) <= 1e-9 * b.skat_p
    ok, U, Vs, eff, se, p, cnt = engine.score_bed_dev(d_bed, int(sum(Ms)))
    raw_all = np.concatenate(genes, axis=1)
    assert np.array_equal(cnt, np.stack([(raw_all == 0).sum(0), (raw_all == 1).sum(0), (raw_all == 2).sum(0), (raw_all < 0).sum(0)], axis=1))
    engine.bed_free(d_bed)
