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


def _run(engine, genes, hard):
    """Run the genes with the blocks registered as hard-call (hard=True) or unknown (general path)."""
    ptrs = [engine.upload_block(G) for G, af in genes]         # upload classifies
    if not hard:
        for p in ptrs:
            engine.forget_block(p)
    engine.set_profiling(True)
    engine.timing(reset=True)
    out = engine.run_blocks(ptrs, [G.shape[1] for G, af in genes], [af for G, af in genes])
    tm = engine.timing(reset=True)
    engine.set_profiling(False)
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


def test_classification(engine):
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
    # an imputed block (fractional means) must take the general path and still match the oracle
    Graw, Gi, afi = synth.make_gene(N, 25, seed=3, missing=0.02)
    out, tm = _run(engine, [(Gi, afi)], True)
    assert tm.genes_hard_call == 0
    rc, a = orc.skat(Gi, afi, X, res, v, 0)
    assert abs(out[0].skat_Q - a.Q) <= 1e-10 * a.Q


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
            assert abs(x - y_) <= 1e-9 * abs(y_) + 1e-300, (G.shape[1], f, x, y_)
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
        assert abs(getattr(a, f) - getattr(b, f)) <= 1e-9 * abs(getattr(b, f)) + 1e-300, f
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
        assert abs(getattr(a, f) - getattr(b, f)) <= 1e-8 * abs(getattr(b, f)) + 1e-300, f


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
    """rvt_submit_gene / _i8 / _bed classify the block they write; hard calls without missing values then run on the
    int8 path and give the oracle's numbers."""
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
    assert len(got) <= 64                     # the last, incomplete group of 16 cannot have been launched
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
        assert abs(x - y_) <= 1e-9 * abs(y_) + 1e-300, (f, x, y_)
