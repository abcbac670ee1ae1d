"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): burden counts / collapsed genotypes bit-exact; Q statistics and p-values
within 1e-6 relative.  p-values additionally get an absolute floor of 1e-14 because both Davies
(1 - qf, acc = 1e-6) and the non-central Liu tail (0.5 + (0.5 - cdf)) are computed by subtraction from 1 in
the reference itself, so their last digits are absolute-, not relative-, accurate.
"""
import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu

REL = 1e-6
ABS_P = 1e-14
# SKAT-O reports 1 - (adaptive integral of ~800 Davies values): the reference's own result carries rounding
# noise of a few hundred ulp of 1.0, so p-values below ~1e-7 only agree to that absolute level.
ABS_SKATO = 5e-13


def close(a, b, rel=REL, abs_=0.0):
    return abs(a - b) <= rel * abs(b) + abs_


@pytest.mark.parametrize("N,M,d,binary", [(1000, 7, 1, 0), (4099, 30, 3, 0), (2500, 50, 3, 1), (777, 16, 2, 0),
                                          (5000, 81, 4, 0), (3001, 33, 1, 1), (640, 1, 3, 0), (9000, 96, 3, 0)])
def test_suffstat_matches_numpy(engine, N, M, d, binary):
    Graw, G, af = synth.make_gene(N, M, seed=N + M, missing=0.01, common=True, mono=True)
    X, y, res, v, s2 = synth.make_null(N, d, binary, seed=1)
    engine.set_null(binary, X, res, v, s2)
    ptr = engine.upload_block(G)
    S, T, u, cs, mn, mx = engine.debug_suffstat(ptr, M)
    engine.free_block(ptr)
    w = v if binary else np.ones(N)
    S0 = (G * w[:, None]).T @ G
    T0 = (G * w[:, None]).T @ X
    u0 = G.T @ res
    tol = 1e-11
    assert np.max(np.abs(S - S0)) <= tol * np.max(np.abs(S0))
    assert np.max(np.abs(T - T0)) <= tol * max(np.max(np.abs(T0)), 1.0)
    assert np.max(np.abs(u - u0)) <= tol * max(np.max(np.abs(G).T @ np.abs(res)), 1e-300)
    assert np.allclose(S, S.T, rtol=0, atol=0)
    # exact column statistics (min/max bit-exact; sums exact up to fp64 re-association of the imputed means)
    assert np.array_equal(mn, G.min(0)) and np.array_equal(mx, G.max(0))
    assert np.max(np.abs(cs - G.sum(0))) <= 1e-12 * N


@pytest.mark.parametrize("N,M,binary,seed", [(1000, 12, 0, 1), (4099, 40, 0, 2), (2049, 64, 1, 3), (513, 3, 0, 4),
                                             (3000, 90, 0, 5)])
def test_collapse_bit_exact(engine, N, M, binary, seed):
    Graw, G, af = synth.make_gene(N, M, seed=seed, missing=0.02, common=True, mono=True, maf_hi=-0.7)
    X, y, res, v, s2 = synth.make_null(N, 2, binary, seed=seed)
    engine.set_null(binary, X, res, v, s2)
    ptr = engine.upload_block(G)
    cmc, zeg, fl, kp = engine.debug_collapse(ptr, M)
    engine.free_block(ptr)
    Gf, fl0, kp0 = orc.flip_poly(G)
    assert np.array_equal(fl, fl0)
    assert np.array_equal(kp, kp0)
    assert np.array_equal(cmc, orc.collapse(Gf, 0))
    assert np.array_equal(zeg, orc.collapse(Gf, 1))


def _check_gene(r, G, af, X, y, res, v, binary, d):
    rc, a = orc.skat(G, af, X, res, v, binary)
    rc2, o = orc.skato(G, af, X, res, v, binary)
    assert r.n_poly == a.n_poly
    if a.n_poly == 0:
        assert r.status & 1
        assert not r.skat_ok and not r.skato_ok
        return
    assert r.skat_ok == 1
    assert close(r.skat_Q, a.Q, 1e-10)
    assert close(r.skat_p, a.pvalue, REL, ABS_P), (r.skat_p, a.pvalue)
    if rc2 == 0:
        assert r.skato_ok == 1
        assert close(r.skato_Q, o.Q, 1e-10)
        assert r.skato_rho == o.rho
        assert close(r.skato_p, o.pvalue, REL, ABS_SKATO), (r.skato_p, o.pvalue)
    else:
        assert r.skato_ok == 0
    if not (binary and d > 1):  # reference behaviour undefined there
        for which, ok, stat, p, nonref in ((0, r.cmc_ok, r.cmc_stat, r.cmc_p, r.cmc_nonref),
                                           (1, r.zeg_ok, r.zeg_stat, r.zeg_p, None)):
            rc3, b = orc.burden(G, X, y, binary, which)
            assert ok == (rc3 == 0)
            if ok:
                # (a statistic that is zero up to rounding: U cancels to almost nothing for a null gene; under a binary trait
                #  the null tile is 42-bit fixed point per column, which moves U by ~1e-12 of its natural scale)
                assert close(stat, b.stat, 1e-9, 1e-11 if binary else 1e-13)
                assert close(p, b.pvalue, REL, ABS_P)
                if nonref is not None:
                    assert nonref == b.nonref_site  # bit-exact count


@pytest.mark.parametrize("binary,d", [(0, 3), (1, 1), (0, 1), (1, 3)])
def test_full_pipeline_vs_oracle(engine, binary, d):
    N = 3000
    rng = np.random.default_rng(100 + binary * 10 + d)
    genes = []
    for g in range(24):
        M = int(rng.integers(1, 60))
        Graw, G, af = synth.make_gene(N, M, seed=1000 * binary + 10 * d + g, missing=0.01 if g % 3 == 0 else 0.0,
                                      common=(g % 4 == 1), mono=(g % 5 == 2), maf_hi=-1.0)
        genes.append((G, af))
    eff = 0.5 * genes[0][0][:, :3].sum(1)
    X, y, res, v, s2 = synth.make_null(N, d, binary, seed=77, G_effect=eff)
    engine.set_null(binary, X, res, v, s2)
    ptrs = [engine.upload_block(G) for G, af in genes]
    out = engine.run_blocks(ptrs, [G.shape[1] for G, af in genes], [af for G, af in genes])
    for p in ptrs:
        engine.free_block(p)
    for r, (G, af) in zip(out, genes):
        _check_gene(r, G, af, X, y, res, v, binary, d)
    assert [r.gene_id for r in out] == list(range(len(genes)))


@pytest.mark.parametrize("M,binary", [(97, 0), (130, 0), (200, 1)])
def test_wide_genes_use_the_panelled_kernel(engine, M, binary):
    """M > 96 goes through gene_suffstat_panel (4 x 4 tile panels) and the global-memory eigen path."""
    N, d = 2500, 2
    Graw, G, af = synth.make_gene(N, M, seed=M, missing=0.01, common=True, mono=True, maf_hi=-1.0)
    narrow = synth.make_gene(N, 20, seed=3)[1:]
    X, y, res, v, s2 = synth.make_null(N, d, binary, seed=M, G_effect=0.4 * G[:, :3].sum(1))
    engine.set_null(binary, X, res, v, s2)
    ptr = engine.upload_block(G)
    S, T, u, cs, mn, mx = engine.debug_suffstat(ptr, M)
    w = v if binary else np.ones(N)
    S0 = (G * w[:, None]).T @ G
    assert np.max(np.abs(S - S0)) <= 1e-11 * np.max(np.abs(S0))
    assert np.max(np.abs(T - (G * w[:, None]).T @ X)) <= 1e-11 * max(np.max(np.abs(S0)), 1.0)
    assert np.array_equal(mn, G.min(0)) and np.array_equal(mx, G.max(0))
    cmc, zeg, fl, kp = engine.debug_collapse(ptr, M)
    Gf, fl0, kp0 = orc.flip_poly(G)
    assert np.array_equal(fl, fl0) and np.array_equal(kp, kp0)
    assert np.array_equal(cmc, orc.collapse(Gf, 0)) and np.array_equal(zeg, orc.collapse(Gf, 1))
    p2 = engine.upload_block(narrow[0])
    out = engine.run_blocks([p2, ptr], [20, M], [narrow[1], af])      # mixed batch: narrow + wide
    engine.free_block(ptr)
    engine.free_block(p2)
    _check_gene(out[1], G, af, X, y, res, v, binary, d)
    _check_gene(out[0], narrow[0], narrow[1], X, y, res, v, binary, d)


def test_streaming_interface_matches_batch(engine):
    N, d = 2000, 2
    genes = [synth.make_gene(N, M, seed=50 + M)[1:] for M in (5, 17, 33)]
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=5)
    engine.set_null(0, X, res, v, s2)
    for i, (G, af) in enumerate(genes):
        engine.submit_gene(100 + i, G, af)
    got = engine.collect()
    assert [r.gene_id for r in got] == [100, 101, 102]
    ptrs = [engine.upload_block(G) for G, af in genes]
    ref = engine.run_blocks(ptrs, [G.shape[1] for G, af in genes], [af for G, af in genes])
    for p in ptrs:
        engine.free_block(p)
    for a, b in zip(got, ref):
        assert a.skat_p == b.skat_p and a.skato_p == b.skato_p and a.cmc_p == b.cmc_p and a.zeg_p == b.zeg_p


def test_determinism(engine):
    N, d = 5000, 3
    G, af = synth.make_gene(N, 45, seed=9)[1:]
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=9)
    engine.set_null(0, X, res, v, s2)
    ptr = engine.upload_block(G)
    a = engine.run_blocks([ptr], [45], [af])[0]
    b = engine.run_blocks([ptr], [45], [af])[0]
    engine.free_block(ptr)
    for f in ("skat_Q", "skat_p", "skato_Q", "skato_p", "cmc_stat", "zeg_stat"):
        assert getattr(a, f) == getattr(b, f)


def test_full_size_properties_binary(engine):
    """BASELINE configs[3] size (binary trait, logistic null, N = 200 000, M = 50): as test_full_size_properties with
    the weighted statistics G'W[G X rr], W = diag(p(1-p)) of the device-fitted logistic null."""
    import hc
    N, M, d = 200000, 50, 2
    rng = np.random.default_rng(20260003)
    maf = 10 ** rng.uniform(np.log10(5e-4), np.log10(5e-2), M)
    G = np.asfortranarray((rng.random((N, M)) < maf).astype(np.float64) + (rng.random((N, M)) < maf))
    af = G.sum(0) / (2.0 * N)
    X = np.column_stack([np.ones(N), rng.normal(size=N)])
    eta = -2.0 + 0.3 * X[:, 1] + 0.1 * G[:, :5].sum(1)
    y = (rng.random(N) < 1.0 / (1.0 + np.exp(-eta))).astype(np.float64)
    rc, beta, p, v = orc.fit_logistic(X, y)
    assert rc == 0
    res = y - p
    gb, _ = engine.fit_null(1, X, y)                         # LogisticRegression::FitLogisticModel on the device
    assert np.allclose(gb, beta, rtol=1e-9, atol=1e-12)
    ptr = engine.upload_block(G)
    S, T, u, cs, mn, mx = engine.debug_suffstat(ptr, M)
    Gw = G * v[:, None]
    assert np.max(np.abs(S - G.T @ Gw)) <= 1e-10 * np.abs(S).max()
    assert np.array_equal(cs, G.sum(0))
    assert np.max(np.abs(T - Gw.T @ X)) <= 1e-10 * N and np.max(np.abs(u - G.T @ res)) <= 1e-10 * N
    r1 = engine.run_blocks([ptr], [M], [af])[0]
    c = (G.astype(np.int64) > 0).sum(1).astype(np.float64)
    bs = hc.burden_sums((c > 0).astype(np.float64), c, X, res, v, 1)
    h, _, _, _ = hc.gene(G, af, X, res, v, 1, 1.0, bstats=bs)
    assert abs(r1.skat_Q - h.skat_Q) <= 1e-10 * h.skat_Q and abs(r1.skat_p - h.skat_p) <= 1e-6 * h.skat_p + ABS_P
    assert r1.skato_rho == h.skato_rho and abs(r1.skato_p - h.skato_p) <= 1e-6 * h.skato_p + ABS_SKATO
    assert r1.cmc_nonref == int((c > 0).sum())
    assert abs(r1.cmc_p - h.cmc_p) <= 1e-7 * h.cmc_p and abs(r1.zeg_p - h.zeg_p) <= 1e-7 * h.zeg_p
    engine.free_block(ptr)


def test_full_size_properties(engine):
    """BASELINE configs[2] size (N = 500 000, M = 50): the oracle's literal SKAT-O needs ~20 s per gene here, so the
    full-size check uses size-independent properties instead:
      * sufficient statistics equal numpy's G'[G X r] (exact for the integer-valued genotype part),
      * the p-values equal those of the SAME algorithms run on the host from numpy's statistics (hc.gene),
      * invariance under a permutation of the samples and under reordering the genes of a batch,
      * burden counts equal a numpy restatement of the collapsers (bit-exact)."""
    import hc
    N, M, d = 500000, 50, 3
    rng = np.random.default_rng(20260002)
    maf = 10 ** rng.uniform(np.log10(5e-4), np.log10(5e-2), M)
    G = np.asfortranarray((rng.random((N, M)) < maf).astype(np.float64) + (rng.random((N, M)) < maf))
    af = G.sum(0) / (2.0 * N)
    X = np.column_stack([np.ones(N), rng.normal(size=N), rng.normal(size=N)])
    y = 0.3 * X[:, 1] - 0.2 * X[:, 2] + rng.normal(size=N) + 0.05 * G[:, :5].sum(1)
    beta = np.linalg.solve(X.T @ X, X.T @ y)
    res = y - X @ beta
    s2 = float(res @ res / N)
    v = np.full(N, s2)
    engine.set_null(0, X, res, v, s2)
    ptr = engine.upload_block(G)
    S, T, u, cs, mn, mx = engine.debug_suffstat(ptr, M)
    assert np.array_equal(S, G.T @ G)                      # integers: exact in any summation order
    assert np.array_equal(cs, G.sum(0))
    assert np.max(np.abs(T - G.T @ X)) <= 1e-10 * N and np.max(np.abs(u - G.T @ res)) <= 1e-10 * N
    r1 = engine.run_blocks([ptr], [M], [af])[0]
    # same algorithms on the host, fed with numpy's statistics
    c = (G.astype(np.int64) > 0).sum(1).astype(np.float64)
    bs = hc.burden_sums((c > 0).astype(np.float64), c, X, res, v, 0)
    h, _, _, _ = hc.gene(G, af, X, res, v, 0, s2, bstats=bs)
    assert abs(r1.skat_Q - h.skat_Q) <= 1e-11 * h.skat_Q and abs(r1.skat_p - h.skat_p) <= 1e-6 * h.skat_p + ABS_P
    assert r1.skato_rho == h.skato_rho and abs(r1.skato_p - h.skato_p) <= 1e-6 * h.skato_p + ABS_SKATO
    assert r1.cmc_nonref == int((c > 0).sum())
    assert abs(r1.cmc_p - h.cmc_p) <= 1e-8 * h.cmc_p and abs(r1.zeg_p - h.zeg_p) <= 1e-8 * h.zeg_p
    # permutation of the samples
    perm = rng.permutation(N)
    engine.set_null(0, X[perm], res[perm], v, s2)
    ptr2 = engine.upload_block(np.asfortranarray(G[perm]))
    r2 = engine.run_blocks([ptr2, ptr2], [M, M], [af, af])
    assert r2[0].skat_p == r2[1].skat_p and r2[0].skato_p == r2[1].skato_p            # order within a batch
    assert abs(r2[0].skat_p - r1.skat_p) <= 1e-6 * r1.skat_p + ABS_P
    assert abs(r2[0].skato_p - r1.skato_p) <= 1e-6 * r1.skato_p + ABS_SKATO
    assert r2[0].cmc_nonref == r1.cmc_nonref and abs(r2[0].cmc_stat - r1.cmc_stat) <= 1e-9 * r1.cmc_stat
    engine.free_block(ptr)
    engine.free_block(ptr2)
