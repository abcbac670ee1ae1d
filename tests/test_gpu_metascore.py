"""GPU parity of the MetaScore single-variant statistics (rvt_score_block, through the C ABI) against the CPU oracle."""
import numpy as np
import pytest

import orc
from test_metacov_cpu import make_case
from test_gpu_metacov import engine_factory  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

REL = 1e-6   # BASELINE.json north_star tolerance for statistics and p-values


def check(o, r):
    assert (r["ok"] == o["ok"]).all()
    k = o["ok"].astype(bool)
    for f in ("U", "V", "effect", "se", "p"):
        scale = max(np.abs(o[f][k]).max(), 1e-300)
        assert np.allclose(r[f][k], o[f][k], rtol=REL, atol=1e-9 * scale), f


@pytest.mark.parametrize("binary", [0, 1])
@pytest.mark.parametrize("N,V,d", [(1500, 7, 1), (1500, 16, 3), (2000, 61, 2), (1203, 200, 5)])
def test_score_block_matches_oracle(engine_factory, binary, N, V, d):
    G, chrom, pos, X, y = make_case(N, V, d, binary, 900 + V + d + 7 * binary)
    if V > 10:
        G[:, 3] = 2.0                                  # monomorphic
        G[:, 9] = np.where(G[:, 9] > 0, 0.123, 0.0)    # non-integer dosage
    eng = engine_factory()
    eng.fit_null(binary, X, y)
    ptr = eng.upload_block(G)
    r = eng.score_block(ptr, V)
    rc, o = orc.metascore(G, X, y, binary)
    assert rc == 0
    check(o, r)
    beta, covb, s2 = eng.null_summary()
    assert np.allclose(beta, o["beta"], rtol=1e-6, atol=1e-9)
    assert np.allclose(covb, o["covb"], rtol=1e-6)
    assert s2 == pytest.approx(o["sigma2"], rel=1e-8)
    eng.free_block(ptr)


def test_score_block_many_slices_and_covariates(engine_factory):
    """More than one launch chunk (> 4096 columns) and the widest covariate set (d = 16: the panelled kernel)."""
    N, d = 640, 16
    G, chrom, pos, X, y = make_case(N, 300, d, 0, 4242)
    G = np.tile(G, (1, 14))[:, :4150].copy(order="F")
    eng = engine_factory()
    eng.fit_null(0, X, y)
    ptr = eng.upload_block(G)
    r = eng.score_block(ptr, G.shape[1])
    rc, o = orc.metascore(G, X, y, 0)
    assert rc == 0
    check(o, r)
    eng.free_block(ptr)


def test_score_block_consistent_with_cov_block(engine_factory):
    """V_STAT is the diagonal of the MetaCov band and U_STAT^2 / V_STAT the printed statistic."""
    N, V, d = 1800, 40, 3
    G, chrom, pos, X, y = make_case(N, V, d, 0, 77)
    eng = engine_factory()
    eng.fit_null(0, X, y)
    ptr = eng.upload_block(G)
    r = eng.score_block(ptr, V)
    cov, xz, zz, poly = eng.cov_block(ptr, V)
    k = r["ok"].astype(bool)
    assert (poly.astype(bool) == k).all()
    assert np.allclose(np.diag(cov)[k], r["V"][k], rtol=1e-8)
    eng.free_block(ptr)


def test_score_block_needs_null(engine_factory):
    import rvtests_amd
    eng = engine_factory()
    with pytest.raises(rvtests_amd.RvtError):
        eng.score_block(1, 4)


@pytest.mark.parametrize("binary", [0, 1])
@pytest.mark.parametrize("n_fam,d", [(40, 1), (60, 3)])
def test_score_block_fam_matches_oracle(engine_factory, n_fam, d, binary):
    """MetaFamQtl / MetaFamBinary: FastLMM score test (+ b scaling, uncentred genotypes for the binary trait) and GLS
    allele frequency of every raw column, and GetNullCovB."""
    import synth
    from test_fam_cpu import make_family_case
    N, K, U, S, X, y = make_family_case(n_fam, d, 190 + d)
    b = 1.0
    if binary:
        y = (y > np.median(y)).astype(float)       # MetaFamBinary fits the LMM to the 0/1 phenotype as it is
    eng = engine_factory()
    eng.set_kinship(U, S)
    nul = eng.fit_fam_null(X, y)
    if binary:
        alpha, b = eng.fam_binary_scale(int((y == 1).sum()), int((y == 0).sum()))
        assert b == pytest.approx(orc.obtain_b(alpha), rel=1e-6)
    onul = orc.FamNull()
    onul.ok = 1
    onul.delta, onul.sigma2 = nul.delta, nul.sigma2_g
    for k in range(d):
        onul.beta[k] = nul.beta[k]
    G = synth.make_gene(N, 37, seed=1234, missing=0.02, common=True, mono=True, maf_hi=-0.8)[1]
    ptr = eng.upload_block(G)
    r = eng.score_block_fam(ptr, G.shape[1], binary)
    tested = 0
    for h in range(G.shape[1]):
        rc, o = orc.fam_burden(G[:, [h]], X, y, U, S, onul, 3 if binary else 2)
        assert r["ok"][h] == (1 if rc == 0 else 0)
        if rc:
            continue
        tested += 1
        assert abs(r["U"][h] - o.U * b) <= 1e-8 * abs(o.U * b) + 1e-12
        assert abs(r["V"][h] - o.V * b * b) <= 1e-8 * o.V * b * b
        assert abs(r["af"][h] - o.af) <= 1e-9 * abs(o.af) + 1e-15
        assert abs(r["p"][h] - o.pvalue) <= 1e-6 * o.pvalue
    assert 5 < tested < G.shape[1]
    rc, covb = orc.fastlmm_covb(X, U, S, nul.delta)
    assert rc == 0
    assert np.allclose(eng.fam_null_summary(d), np.diag(covb), rtol=1e-8)
    eng.free_block(ptr)


def test_score_block_fam_chunks_beyond_one_block(engine_factory):
    """More raw columns than RVT_MAX_VARIANTS: rvt_score_block_fam walks the block in pieces; every column equals what a
    call on a small block that holds only its neighbourhood returns."""
    import synth
    from test_fam_cpu import make_family_case
    N, K, U, S, X, y = make_family_case(40, 2, 77)
    eng = engine_factory()
    eng.set_kinship(U, S)
    eng.fit_fam_null(X, y)
    V = 1100
    G = synth.make_gene(N, V, seed=99, missing=0.02, common=True, mono=True, maf_hi=-0.7)[1]
    ptr = eng.upload_block(G)
    whole = eng.score_block_fam(ptr, V)
    eng.free_block(ptr)
    for lo, hi in ((0, 40), (1000, 1060), (1060, 1100)):
        p2 = eng.upload_block(np.asfortranarray(G[:, lo:hi]))
        part = eng.score_block_fam(p2, hi - lo)
        eng.free_block(p2)
        assert (part["ok"] == whole["ok"][lo:hi]).all() and part["ok"].sum() > 5
        for f in ("U", "V", "af", "p"):
            assert np.allclose(part[f], whole[f][lo:hi], rtol=1e-9, atol=1e-12), f


@pytest.mark.parametrize("binary", [0, 1])
@pytest.mark.parametrize("mixed", [False, True])
def test_score_block_hard_call_slices(engine_factory, mixed, binary, monkeypatch):
    """Blocks filled column by column (the adapter's way): the per-column content flags send all-hard-call slices
    through the int8 kernel and the others through the fp64 kernel; both must give the oracle's numbers, and the same
    as the fp64 kernel alone (RVT_HARDCALL=0)."""
    N, V, d = 3000, 150, 3
    G, chrom, pos, X, y = make_case(N, V, d, binary, 31337)
    G = np.rint(G)                                     # hard calls
    G[:, 5] = 1.0                                      # monomorphic, non-zero
    if mixed:
        G[:, 40] = np.where(G[:, 40] > 0, 0.5, 0.0)    # one dosage column: its slice takes the general kernel
        G[17, 99] = 1.0 / 3.0
    rc, o = orc.metascore(G, X, y, binary)
    assert rc == 0
    out = []
    for hc in ("1", "0"):
        monkeypatch.setenv("RVT_HARDCALL", hc)
        eng = engine_factory()
        eng.fit_null(binary, X, y)
        eng.set_profiling(True)
        ptr = eng.alloc_block(V)
        for j in range(0, V, 7):                       # ragged column-wise fill
            eng.upload_columns(ptr, j, G[:, j:j + 7])
        r = eng.score_block(ptr, V)
        check(o, r)
        t = eng.timing()
        out.append((r, t))
        eng.free_block(ptr)
    (r1, t1), (r0, t0) = out
    assert t1.n_suffstat_hc_launches >= 1 and t0.n_suffstat_hc_launches == 0
    for f in ("U", "V", "p"):
        k = o["ok"].astype(bool)
        assert np.allclose(r1[f][k], r0[f][k], rtol=1e-8 if binary else 1e-11, atol=0)   # (binary: digit planes of v)


@pytest.mark.parametrize("N,V,d", [(4099, 70, 2), (10007, 1000, 3), (3001, 8300, 1)])
def test_score_of_resident_bed_rows_matches_the_oracle(engine_factory, monkeypatch, N, V, d):
    """rvt_score_bed_dev: the single-variant score test straight from the 2-bit rows of a resident .bed matrix (missing calls
    imputed to the column mean, as DataConsolidator does) — against the oracle's MetaScore on the imputed matrix, against
    rvt_score_block on the same matrix as doubles, and the genotype counts against numpy.  V = 8 300: more than one chunk of
    256 slices; V not a multiple of 32: a short last slice."""
    rng = np.random.default_rng(N + V)
    maf = 10 ** rng.uniform(-2.5, -0.4, V)
    raw = rng.binomial(2, maf, size=(N, V)).astype(np.float64)
    raw[rng.random((N, V)) < 0.01] = -9.0
    raw[:, 3] = np.where(raw[:, 3] < 0, -9.0, 2.0)            # monomorphic apart from its missing calls
    raw[:, 5] = 0.0
    if V > 40:
        raw[:, 40] = -9.0                                     # nothing but missing calls
    X = np.column_stack([np.ones(N)] + [rng.normal(size=N) for _ in range(d - 1)])
    y = X @ rng.normal(size=d) + 0.2 * np.maximum(raw[:, 7], 0) + rng.normal(size=N)
    monkeypatch.setenv("RVT_POISON", "255")                   # fresh work spaces hold 0xff bytes: nothing may be read unwritten
    eng = engine_factory()
    eng.fit_null(0, np.asfortranarray(X), y)
    d_bed = eng.bed_alloc(V + 2)
    cb = (N + 3) // 4
    eng.bed_upload(d_bed, 0, np.zeros((2, cb), dtype=np.uint8))
    eng.bed_upload(d_bed, 2, eng.pack_bed(raw))
    ok, U, Vs, eff, se, p, cnt = eng.score_bed_dev(d_bed + 2 * cb, V)
    G = orc.impute_mean(np.asfortranarray(raw))
    rc, o = orc.metascore(G, np.asfortranarray(X), y, 0)
    assert rc == 0
    check(o, dict(ok=ok, U=U, V=Vs, effect=eff, se=se, p=p))
    want_cnt = np.stack([(raw == 0).sum(0), (raw == 1).sum(0), (raw == 2).sum(0), (raw < 0).sum(0)], axis=1)
    assert np.array_equal(cnt, want_cnt)
    ptr = eng.upload_block(G)
    r = eng.score_block(ptr, V)
    k = ok.astype(bool)
    assert (r["ok"] == ok).all()
    for a, b in ((U, r["U"]), (Vs, r["V"]), (eff, r["effect"]), (se, r["se"]), (p, r["p"])):
        assert np.allclose(a[k], b[k], rtol=1e-9, atol=1e-12 * max(np.abs(b[k]).max(), 1e-300))
    eng.free_block(ptr)
    eng.bed_free(d_bed)


def test_score_of_resident_rows_after_gene_batches_used_the_work_space(engine_factory):
    """The score path of resident rows runs ONE streaming kernel and must leave everything the flag kernel reads — also the
    per-wave-part "an entry outside the codes" word, which lives in a work space that gene batches before it have written: a
    stale bit there hands a slice of packed rows to the fp64 kernel (round 6: a memory fault in the bench run, found by it).
    2 048 genes first, then the score test of the same matrix — equal to the score test of a fresh context."""
    N, V = 20011, 4096
    rng = np.random.default_rng(77)
    raw = rng.binomial(2, 10 ** rng.uniform(-2.3, -0.5, V), size=(N, V)).astype(np.float64)
    raw[rng.random((N, V)) < 0.02] = -9.0
    X = np.column_stack([np.ones(N), rng.normal(size=N)])
    y = rng.normal(size=N) + 0.3 * X[:, 1]
    outs = []
    for dirty in (True, False):
        eng = engine_factory()
        eng.fit_null(0, np.asfortranarray(X), y)
        cb = (N + 3) // 4
        d_bed = eng.bed_alloc(V)
        eng.bed_upload(d_bed, 0, eng.pack_bed(raw))
        if dirty:
            Ms = rng.integers(20, 81, 2048)
            first = rng.integers(0, V - 80, 2048)
            for g0 in range(0, 2048, 64):
                eng.submit_genes_bed_dev(list(range(g0, g0 + 64)), [d_bed + int(f) * cb for f in first[g0:g0 + 64]], Ms[g0:g0 + 64])
                eng.collect_ready()
            assert len(eng.collect()) > 0
        outs.append(eng.score_bed_dev(d_bed, V))
        eng.bed_free(d_bed)
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    assert outs[0][0].sum() > V // 2
