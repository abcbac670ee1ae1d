"""GPU: the lattice-dosage sufficient-statistics path (rvtests_amd/csrc/suffstat_lat.hip.h — dosages K / den, two 7-bit
digits of K on the int8 matrix cores, burden collapse in the same pass) against the general fp64 path on the same
blocks, against exact integer arithmetic, and against the oracle; and that blocks which are not on the stated lattice
are handed back to the fp64 kernel with the same records."""
import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu

FIELDS = ("skat_Q", "skat_p", "skato_Q", "skato_p", "skato_rho", "cmc_U", "cmc_V", "cmc_stat", "cmc_p", "zeg_U",
          "zeg_V", "zeg_stat", "zeg_p")


def _dosage_K(N, M, seed, den=1000, common_col=None, ones_col=None, const_col=None, maf_hi=-1.0):
    """Imputed dosages of rare variants as integers K (dosage = K / den): a hard call blurred by the imputation."""
    rng = np.random.default_rng(seed)
    maf = 10 ** rng.uniform(-3.0, maf_hi, M)
    H = rng.binomial(2, maf, size=(N, M))
    blur = np.rint(np.abs(rng.normal(0.0, 0.04, size=(N, M))) * den).astype(np.int64)
    blur[rng.random((N, M)) < 0.7] = 0                      # most entries are printed as 0 / 1.000 / 2.000
    K = np.where(H == 2, 2 * den - blur, np.where(H == 1, den + blur * rng.choice([-1, 1], size=(N, M)), blur))
    if common_col is not None and M > common_col:           # a column whose sum exceeds N: flipped by the reference
        Hc = rng.binomial(2, 0.85, size=N)
        K[:, common_col] = np.clip(Hc * den - blur[:, common_col] * (Hc > 0), 0, 2 * den)
    if ones_col is not None and M > ones_col:               # many values exactly 1.0 (the boundary of (int)g' > 0)
        K[:, ones_col] = np.where(rng.random(N) < 0.5, den, K[:, ones_col])
    if const_col is not None and M > const_col:
        K[:, const_col] = den // 2                          # monomorphic, not an integer
    return np.clip(K, 0, 2 * den).astype(np.int64)


def _gene(K, den):
    G = np.asfortranarray(K.astype(np.float64) / float(den))     # the double nearest to K / den, as strtod gives it
    return G, G.sum(0) / (2.0 * G.shape[0])


def _run(engine, genes, mode, den=1000):
    """mode 'lat': dosages on the stated lattice; 'gen': the engine confined to the fp64 kernel."""
    ptrs = [engine.upload_block(G) for G, af in genes]
    engine.set_content_hint(0)
    engine.set_dosage_lattice(den if mode == "lat" else 0)
    engine.set_hardcall(mode == "lat")
    engine.set_profiling(True)
    engine.timing(reset=True)
    try:
        out = engine.run_blocks(ptrs, [G.shape[1] for G, af in genes], [af for G, af in genes])
        tm = engine.timing(reset=True)
    finally:
        engine.set_profiling(False)
        engine.set_hardcall(True)
        engine.set_dosage_lattice(0)
        engine.set_content_hint(-1)
    for p in ptrs:
        engine.free_block(p)
    return out, tm


@pytest.mark.parametrize("N,d,den", [(3000, 3, 1000), (4099, 1, 1000), (10007, 2, 100), (2500, 2, 255), (1777, 1, 2048)])
def test_lattice_path_equals_general_path_and_oracle(engine, N, d, den):
    Ks = [_dosage_K(N, M, seed=13 * M + d, den=den, common_col=(2 if M % 3 == 0 else None),
                    ones_col=(4 if M % 2 == 0 else None), const_col=(1 if M % 5 == 2 else None))
          for M in (1, 7, 16, 17, 30, 33, 48, 50, 64, 65, 80, 81, 96)]
    genes = [_gene(K, den) for K in Ks]
    eff = 0.4 * genes[4][0][:, :3].sum(1)
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=5, G_effect=eff)
    engine.set_null(0, X, res, v, s2)
    lat, tm_lat = _run(engine, genes, "lat", den)
    gen, tm_gen = _run(engine, genes, "gen")
    n_lat = sum(1 for G, af in genes if G.shape[1] <= 80)
    assert tm_lat.genes == len(genes) and tm_lat.genes_hard_call == n_lat and tm_lat.genes_handed_back == 0
    assert tm_gen.genes_hard_call == 0
    for a, b, (G, af) in zip(lat, gen, genes):
        assert a.n_poly == b.n_poly and a.cmc_nonref == b.cmc_nonref and a.status == b.status
        for f in FIELDS:
            x, y_ = getattr(a, f), getattr(b, f)
            assert abs(x - y_) <= 1e-10 * abs(y_) + 1e-300, (f, x, y_)
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


@pytest.mark.parametrize("N,den", [(5003, 1000), (777, 8), (64, 1000), (50, 100)])
def test_integer_part_is_exact(engine, N, den):
    """G'G = K'K / den^2 with K'K formed in integers: bit-identical to the exact integer matrix divided once; column sums
    likewise; min / max are the doubles themselves."""
    d = 2
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=3)
    engine.set_null(0, X, res, v, s2)
    engine.set_content_hint(0)
    engine.set_dosage_lattice(den)
    try:
        for M in (5, 16, 40, 64, 80):
            K = _dosage_K(N, M, seed=M, den=den, common_col=3, ones_col=1)
            G, af = _gene(K, den)
            p = engine.upload_block(G)
            S, T, u, cs, mn, mx = engine.debug_suffstat(p, M)
            engine.free_block(p)
            exact = (K.T @ K).astype(np.float64) / float(den * den)
            assert np.array_equal(np.triu(S), np.triu(exact))
            assert np.array_equal(cs, K.sum(0).astype(np.float64) / float(den))
            assert np.array_equal(mn, G.min(0)) and np.array_equal(mx, G.max(0))
            assert np.allclose(T, G.T @ X, rtol=1e-12, atol=1e-9)
    finally:
        engine.set_dosage_lattice(0)
        engine.set_content_hint(-1)


def test_blocks_off_the_lattice_are_handed_back(engine):
    """Mean-imputed entries, another number of decimals, arbitrary doubles, values outside [0, 2], the neighbour doubles
    of a lattice point (1 + 2^-52 and 1 - 2^-53 are not the doubles nearest to 1000 / 1000 — the test is |g den - K| <=
    K 2^-53 — except that 1 - 2^-53 sits exactly AT the bound): each such gene is computed by the fp64 kernel in the same
    call — the records of the confined run; gene 6, with lower neighbours only, stays on the lattice kernel."""
    N, d, den = 6000, 2, 1000
    rng = np.random.default_rng(11)
    base = [_gene(_dosage_K(N, M, seed=100 + M, den=den, common_col=2), den) for M in (12, 30, 50, 70, 20, 44, 44)]
    genes = [(G.copy(order="F"), af.copy()) for G, af in base]
    G = genes[0][0]
    miss = rng.random(N) < 0.02
    G[miss, 3] = G[~miss, 3].mean()                              # imputeGenotypeToMean's value
    genes[1][0][rng.integers(N), 5] = 0.12345                    # five decimals
    genes[2][0][:, :] = rng.uniform(0, 2, size=genes[2][0].shape)
    genes[3][0][17, 60] = 2.5
    # genes[4]: untouched
    G5 = genes[5][0]
    G5[G5[:, 7] == 1.0, 7] = np.nextafter(1.0, 0.0)              # (int)g = 0 where the lattice point says 1
    G5[G5[:, 2] == 1.0, 2] = np.nextafter(1.0, 2.0)              # column 2 is flipped: (int)(2 - g) = 0
    G6 = genes[6][0]
    G6[G6[:, 7] == 1.0, 7] = np.nextafter(1.0, 0.0)              # at the bound: passes, counted as (int)g = 0
    genes = [(g, g.sum(0) / (2.0 * N)) for g, _ in genes]
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=9, G_effect=0.3 * base[4][0][:, :4].sum(1))
    engine.set_null(0, X, res, v, s2)
    lat, tm = _run(engine, genes, "lat", den)
    gen, _ = _run(engine, genes, "gen")
    assert tm.genes_hard_call == 7 and tm.genes_handed_back == 5
    for k, (a, b) in enumerate(zip(lat, gen)):
        assert a.status == b.status and a.n_poly == b.n_poly and a.cmc_nonref == b.cmc_nonref
        for f in FIELDS:
            x, y_ = getattr(a, f), getattr(b, f)
            if k in (0, 1, 2, 3, 5) and not f.startswith(("cmc", "zeg")):
                assert x == y_, (k, f)                           # the same kernel computed it
            elif k == 2 and f.startswith("cmc"):
                continue       # every sample counts: the CMC genotype is the constant 1, U = sum of the residuals = rounding
            else:
                assert abs(x - y_) <= 1e-10 * abs(y_) + 1e-300, (k, f, x, y_)
    for k in (5, 6):
        rc, c = orc.burden(genes[k][0], X, y, 0, 0)
        assert lat[k].cmc_nonref == c.nonref_site
        rc, z = orc.burden(genes[k][0], X, y, 0, 1)
        assert abs(lat[k].zeg_stat - z.stat) <= 1e-9 * z.stat


def test_burden_collapse_on_dosages(engine):
    """(int)g' > 0 on dosages: g >= 1 counts for an unflipped column, g <= 1 for a flipped one; a wrong allele frequency
    (flip predicted wrongly) and a monomorphic column that would count are redone by the fallback — all as the oracle."""
    N, d, den = 5003, 2, 1000
    K = _dosage_K(N, 35, seed=2, den=den, common_col=7, ones_col=9)
    K[:, 4] = den                                                # monomorphic 1.0: never counts in the reference
    G, af = _gene(K, den)
    wrong = af.copy()
    wrong[7] = 0.01
    wrong[3] = 0.9
    genes = [(G, af), (G, wrong), _gene(_dosage_K(N, 12, seed=3, den=den), den)]
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=8, G_effect=0.3 * genes[2][0][:, :4].sum(1))
    engine.set_null(0, X, res, v, s2)
    out, tm = _run(engine, genes, "lat", den)
    assert tm.genes_hard_call == 3 and tm.genes_handed_back == 0
    for r, (Gg, a_) in zip(out, genes):
        for which, ok, stat, p in ((0, r.cmc_ok, r.cmc_stat, r.cmc_p), (1, r.zeg_ok, r.zeg_stat, r.zeg_p)):
            rc, b = orc.burden(Gg, X, y, 0, which)
            assert ok == (rc == 0)
            if ok:
                assert abs(stat - b.stat) <= 1e-9 * b.stat + 1e-13 and abs(p - b.pvalue) <= 1e-6 * b.pvalue + 1e-14
        rc, c = orc.burden(Gg, X, y, 0, 0)
        assert r.cmc_nonref == c.nonref_site


def test_lattice_needs_the_hint_and_a_valid_denominator(engine):
    import rvtests_amd
    N, den = 3000, 1000
    X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=4)
    engine.set_null(0, X, res, v, s2)
    with pytest.raises(rvtests_amd.RvtError):
        engine.set_dosage_lattice(4096)
    with pytest.raises(rvtests_amd.RvtError):
        engine.set_dosage_lattice(-1)
    G, af = _gene(_dosage_K(N, 20, seed=5, den=den), den)
    p = engine.upload_block(G)
    engine.set_profiling(True)
    try:
        engine.set_dosage_lattice(den)                           # no hint: the block starts on the hard-call kernel
        engine.timing(reset=True)
        (a,) = engine.run_blocks([p], [20], [af])
        tm = engine.timing(reset=True)
        assert tm.genes_hard_call == 1 and tm.genes_handed_back == 1
        engine.set_content_hint(0)
        (b,) = engine.run_blocks([p], [20], [af])
        tm = engine.timing(reset=True)
        assert tm.genes_hard_call == 1 and tm.genes_handed_back == 0
        assert abs(a.skat_p - b.skat_p) <= 1e-10 * b.skat_p and a.cmc_nonref == b.cmc_nonref
    finally:
        engine.set_profiling(False)
        engine.set_dosage_lattice(0)
        engine.set_content_hint(-1)
        engine.free_block(p)


def test_binary_trait_stays_on_the_fp64_kernel(engine):
    N, den = 4000, 1000
    X, y, res, v, s2 = synth.make_null(N, 2, 1, seed=6)
    engine.set_null(1, X, res, v, s2)
    genes = [_gene(_dosage_K(N, M, seed=M, den=den), den) for M in (10, 40)]
    lat, tm = _run(engine, genes, "lat", den)
    gen, _ = _run(engine, genes, "gen")
    assert tm.genes_hard_call == 0
    for a, b in zip(lat, gen):
        for f in FIELDS:
            assert getattr(a, f) == getattr(b, f), f


def test_streamed_dosage_genes_take_the_lattice_kernel(engine):
    """rvt_submit_gene (imputed doubles + allele frequencies) under the dosage hint, a few batches deep."""
    N, d, den = 3500, 2, 1000
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=12)
    engine.set_null(0, X, res, v, s2)
    rng = np.random.default_rng(1)
    genes = [_gene(_dosage_K(N, int(rng.integers(2, 81)), seed=300 + g, den=den, common_col=1), den) for g in range(70)]
    engine.set_content_hint(0)
    engine.set_dosage_lattice(den)
    engine.set_profiling(True)
    engine.timing(reset=True)
    try:
        for g, (G, af) in enumerate(genes):
            engine.submit_gene(g, G, af)
        got = engine.collect()
        tm = engine.timing(reset=True)
    finally:
        engine.set_profiling(False)
        engine.set_dosage_lattice(0)
    assert tm.genes_hard_call == 70 and tm.genes_handed_back == 0
    try:
        for g, (G, af) in enumerate(genes):
            engine.submit_gene(g, G, af)
        want = engine.collect()
    finally:
        engine.set_content_hint(-1)
    assert [r.gene_id for r in got] == list(range(70))
    for a, b in zip(got, want):
        assert a.n_poly == b.n_poly and a.cmc_nonref == b.cmc_nonref
        for f in FIELDS:
            assert abs(getattr(a, f) - getattr(b, f)) <= 1e-10 * abs(getattr(b, f)) + 1e-300, f


def test_vcf_dosage_text_takes_the_lattice_kernel():
    """`--dosage DS` through the engine's own VCF text decoder: complete three-decimal fields run on the lattice kernel when
    the adapter has stated the lattice ("." is atof's 0.0 there, not a missing call); a gene with a five-decimal field is
    handed back.  Records as without the statement."""
    import rvtests_amd
    eng = rvtests_amd.Engine(0)
    try:
        N, d, n_file = 3000, 2, 3100
        rng = np.random.default_rng(5)
        rows = np.full(n_file, -1, dtype=np.int32)
        keep = rng.choice(n_file, N, replace=False)
        rows[keep] = rng.permutation(N)
        X, y, res, v, s2 = synth.make_null(N, d, 0, seed=14)
        eng.fit_null(0, X, y)
        eng.vcf_set_samples(rows)
        eng.vcf_set_filters(0, 0, 0, 0)
        eng.vcf_set_dosage(True)

        def record(pos, odd):
            vals = np.where(rng.random(n_file) < 0.03, rng.uniform(0.3, 2.0, n_file), rng.uniform(0, 0.08, n_file))
            cols = [b"0/0:%.3f" % x for x in vals]
            cols[int(keep[1])] = b"./.:."
            if odd:
                cols[int(keep[0])] = b"0/1:0.12345"
            head = b"\t".join([b"1", b"%d" % pos, b".", b"A", b"G", b"50", b"PASS", b".", b"GT:DS"])
            return head + b"\t" + b"\t".join(cols)

        genes = [[record(10 * g + j, g == 2) for j in range(9 + 4 * g)] for g in range(4)]
        runs = {}
        for den in (0, 1000):
            eng.set_dosage_lattice(den)
            eng.set_profiling(True)
            eng.timing(reset=True)
            for g, lines in enumerate(genes):
                eng.submit_gene_vcf_dosage(g, lines, b"DS")
            runs[den] = (eng.collect(), eng.timing(reset=True))
            eng.set_profiling(False)
        assert runs[0][1].genes_hard_call == 0
        assert runs[1000][1].genes_hard_call == 4 and runs[1000][1].genes_handed_back == 1
        for a, b in zip(runs[1000][0], runs[0][0]):
            assert a.gene_id == b.gene_id and a.n_poly == b.n_poly and a.cmc_nonref == b.cmc_nonref and a.status == b.status
            for f in FIELDS:
                assert abs(getattr(a, f) - getattr(b, f)) <= 1e-10 * abs(getattr(b, f)) + 1e-300, f
    finally:
        eng.close()


def test_group_members_take_lattice_dosages(engine):
    """rvt_group_set_content: the hint and the lattice on every member of a device group (two contexts on one GPU); the
    records equal the single-context ones."""
    import rvtests_amd
    N, d, den = 2501, 2, 1000
    rng = np.random.default_rng(8)
    genes = [_gene(_dosage_K(N, int(rng.integers(1, 81)), seed=900 + g, den=den, common_col=1), den) for g in range(70)]
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=2, G_effect=0.4 * genes[3][0][:, :2].sum(1))
    grp = rvtests_amd.Group([0, 0])
    try:
        grp.fit_null(0, X, y)
        grp.set_content(0, den)
        with pytest.raises(rvtests_amd.RvtError):
            grp.set_content(0, 5000)
        grp.set_content(0, den)
        for g, (G, af) in enumerate(genes):
            grp.submit_gene(g, G, af)
        got = grp.collect()
    finally:
        grp.close()
    engine.fit_null(0, X, y)
    engine.set_content_hint(0)
    engine.set_dosage_lattice(den)
    try:
        for g, (G, af) in enumerate(genes):
            engine.submit_gene(g, G, af)
        ref = engine.collect()
    finally:
        engine.set_dosage_lattice(0)
        engine.set_content_hint(-1)
    assert [r.gene_id for r in got] == list(range(70))
    for a, b in zip(got, ref):
        for f in ("skat_Q", "skat_p", "skato_Q", "skato_p", "cmc_p", "zeg_p", "cmc_nonref", "n_poly", "status"):
            assert getattr(a, f) == getattr(b, f), f


def test_full_size_and_the_operand_bound():
    """BASELINE's N = 500 000 (exact integers again), and N = 1.1 million with every entry at the largest value the widest
    lattice allows (K = 4096 at den = 2048): the wave-parts then hold the 255 operands the 32-bit tiles are proved for."""
    import rvtests_amd
    eng = rvtests_amd.Engine(0)
    try:
        for N, M, den, worst in ((500_000, 50, 1000, False), (1_100_000, 16, 2048, True)):
            X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=3)
            eng.set_null(0, X, res, v, s2)
            eng.set_content_hint(0)
            eng.set_dosage_lattice(den)
            if worst:
                K = np.full((N, M), 2 * den, dtype=np.int64)
                K[::7, ::3] = 2 * den - 1            # (not monomorphic; low digit 127 next to high digit 31)
            else:
                K = _dosage_K(N, M, seed=77, den=den, common_col=3)
            G, af = _gene(K, den)
            p = eng.upload_block(G)
            S, T, u, cs, mn, mx = eng.debug_suffstat(p, M)
            eng.set_profiling(True)
            eng.timing(reset=True)
            (r,) = eng.run_blocks([p], [M], [af])
            tm = eng.timing(reset=True)
            eng.set_profiling(False)
            eng.free_block(p)
            assert tm.genes_hard_call == 1 and tm.genes_handed_back == 0
            exact = (K.T @ K).astype(np.float64) / float(den * den)
            assert np.array_equal(np.triu(S), np.triu(exact))
            assert np.array_equal(cs, K.sum(0).astype(np.float64) / float(den))
            assert r.n_poly == (M if not worst else len(range(0, M, 3)))
    finally:
        eng.set_dosage_lattice(0)
        eng.set_content_hint(-1)
        eng.close()
