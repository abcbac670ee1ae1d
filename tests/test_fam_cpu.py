"""FastLMM null + FamSKAT oracle (oracle/orc_fam.cpp) against an independent numpy statement of the reference's
formulas (regression/FastLMM.cpp:28-142,283-346,402-443; regression/FamSkat.cpp:34-138)."""
import numpy as np
import pytest

import orc
import synth


def make_family_case(n_fam, d, seed, h2=0.4):
    """Nuclear families of 4 (SURVEY config 5): kinship blocks, U/S by eigendecomposition, phenotype with a
    family random effect."""
    rng = np.random.default_rng(seed)
    N = 4 * n_fam
    blk = np.array([[1, 0, .5, .5], [0, 1, .5, .5], [.5, .5, 1, .5], [.5, .5, .5, 1]])
    K = np.kron(np.eye(n_fam), blk)
    S, U = np.linalg.eigh(K)
    U = U.astype(np.float32).astype(np.float64)      # EigenMatrix holds floats
    S = S.astype(np.float32).astype(np.float64)
    X = np.column_stack([np.ones(N)] + [rng.standard_normal(N) for _ in range(d - 1)])
    L = np.linalg.cholesky(K + 1e-9 * np.eye(N))
    y = X @ rng.standard_normal(d) * 0.3 + np.sqrt(h2) * (L @ rng.standard_normal(N)) + \
        np.sqrt(1 - h2) * rng.standard_normal(N)
    return N, K, U, S, X, y


def loglik(delta, ux, uy, lam):
    D = 1.0 / np.abs(lam + delta)
    A = ux.T @ (ux * D[:, None])
    b = ux.T @ (uy * D)
    beta = np.linalg.solve(A, b)
    r = uy - ux @ beta
    s2 = np.sum(r * r / (lam + delta)) / len(uy)
    n = len(uy)
    return -0.5 * (n * np.log(2 * np.pi) + np.sum(np.log(np.abs(lam + delta))) + n + n * np.log(s2)), beta, s2


@pytest.mark.parametrize("d", [1, 3])
def test_fastlmm_null_is_the_grid_plus_brent_optimum(d):
    N, K, U, S, X, y = make_family_case(40, d, 3 + d)
    rc, nul = orc.fastlmm_null(X, y, U, S)
    assert rc == 0 and nul.ok
    ux, uy, lam = U.T @ X, U.T @ y, np.abs(S)
    grid = np.exp(-10 + 0.2 * np.arange(101))
    lls = np.array([loglik(t, ux, uy, lam)[0] for t in grid])
    assert nul.max_index == int(np.argmax(lls))
    assert 0 < nul.max_index < 100 and nul.brent_evals >= 4
    lo, hi = grid[nul.max_index - 1], grid[nul.max_index + 1]
    assert lo < nul.delta < hi
    fine = np.linspace(lo, hi, 4001)
    best = fine[int(np.argmax([loglik(t, ux, uy, lam)[0] for t in fine]))]
    assert abs(nul.delta - best) < 2e-3 + 1e-3 * best          # Brent stops on an ABSOLUTE 1e-3 bracket
    # beta / sigma2 belong to the LAST point Brent evaluated, which lies inside the final bracket around delta
    ll, beta, s2 = loglik(nul.delta, ux, uy, lam)
    assert np.allclose(nul.beta[:d], beta, rtol=5e-3, atol=5e-4)
    assert abs(nul.sigma2 - s2) < 5e-3 * s2


def test_famskat_matches_numpy_literal():
    N, K, U, S, X, y = make_family_case(30, 2, 11)
    rc, nul = orc.fastlmm_null(X, y, U, S)
    assert rc == 0
    _, G, af = synth.make_gene(N, 12, seed=5, missing=0.02, common=True, mono=True)
    rc, out = orc.famskat(G, X, y, U, S, nul)
    assert rc == 0 and out.fit_ok
    # independent statement
    Gf, fl, kp = orc.flip_poly(G)
    m = Gf.shape[1]
    assert out.n_poly == m
    beta = np.array(nul.beta[:2])
    Sigma = (U * (S + nul.delta)) @ U.T * nul.sigma2
    Sinv = (U / (S + nul.delta)) @ U.T / nul.sigma2
    P0 = Sigma - X @ np.linalg.inv(X.T @ Sinv @ X) @ X.T
    u1 = U.sum(0)
    denom = np.sum(u1 * u1 / np.abs(S))
    alpha = (u1 / np.abs(S)) @ U.T
    afs = 0.5 * (alpha @ Gf) / denom
    from scipy.stats import beta as beta_dist
    w = beta_dist.pdf(afs, 1, 25)
    wg = w[:, None] * Gf.T
    Q = np.sum((wg @ (Sinv @ (y - X @ beta))) ** 2)
    ev = np.linalg.eigvalsh(wg @ P0 @ wg.T)[::-1]
    ev = ev[ev > 1e-30]
    assert abs(out.Q - Q) < 1e-9 * Q
    assert out.n_lambda == len(ev)
    got = np.array(out.lambda_[: out.n_lambda])
    assert np.allclose(got, ev, rtol=1e-8, atol=1e-10 * ev[0])
    p = orc.davies(ev, Q)
    assert abs(out.pvalue - p) <= 1e-6 * p + 1e-12


def test_metacov_fam_matches_numpy():
    N, K, U, S, X, y = make_family_case(30, 2, 17)
    rc, nul = orc.fastlmm_null(X, y, U, S)
    assert rc == 0
    _, G, af = synth.make_gene(N, 15, seed=8, missing=0.02, common=True, mono=True)
    rng = np.random.default_rng(2)
    pos = np.cumsum(rng.integers(1, 300, 15)).astype(np.int32)
    chrom = np.ones(15, dtype=np.int32)
    rc, kept, cov, row_end, xz, zz = orc.metacov_fam(G, chrom, pos, X, U, S, nul, 900)
    assert rc == 0
    w = 1.0 / (np.abs(S) + nul.delta)
    Gt = U.T @ (G - G.mean(0))
    ux = U.T @ X
    XX = Gt.T @ (Gt * w[:, None]) / nul.sigma2
    XZ = Gt.T @ (ux * w[:, None]) / nul.sigma2
    ZZ = ux.T @ (ux * w[:, None]) / nul.sigma2
    val = XX - XZ @ np.linalg.inv(ZZ) @ XZ.T
    k2 = np.array([len(np.unique(G[:, j])) > 1 for j in range(15)])
    assert (kept.astype(bool) == k2).all()
    m = ~np.isnan(cov)
    assert m.sum() > 15
    assert np.abs(cov[m] - val[m]).max() < 1e-9 * np.abs(val[m]).max()
    assert np.allclose(zz, ZZ, rtol=1e-9)
    assert np.allclose(xz[k2], XZ[k2], rtol=1e-8, atol=1e-9 * np.abs(XZ).max())


def test_brent_restatement_matches_gsl_fixture():
    """The oracle's Brent minimiser against GSL 1.16 itself (tests/golden/gsl_brent.json, generated from the library
    the reference vendors): same status, same minimum, same number of evaluations, same last abscissa."""
    import ctypes as C
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gsl_brent.json")))
    L = orc.lib()
    L.orc_brent_builtin.restype = C.c_int
    L.orc_brent_builtin.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double),
                                    C.POINTER(C.c_int), C.POINTER(C.c_double)]
    assert len(fx["cases"]) >= 20
    seen_fail = False
    for c in fx["cases"]:
        xmin, n, last = C.c_double(0), C.c_int(0), C.c_double(0)
        rc = L.orc_brent_builtin(c["id"], c["a"], c["start"], c["lb"], c["ub"], C.byref(xmin), C.byref(n),
                                 C.byref(last))
        assert rc == c["rc"], c
        assert n.value == c["evals"], c
        if rc == 0:
            assert xmin.value == c["xmin"] and last.value == c["last_x"], c      # bit-identical trajectories
        else:
            seen_fail = True
    assert seen_fail


def test_obtain_b_matches_scipy_quad():
    from scipy.integrate import quad
    for alpha in (-3.0, -0.4054651, 0.0, 1.2, 4.0):
        f = lambda x: (lambda t: t / (1 + t) ** 2)(np.exp(alpha + x)) * np.exp(-x * x / 2) / np.sqrt(2 * 3.1415926535897) \
            if alpha + x < 700 else 0.0
        want = quad(f, -40, 40, epsabs=0, epsrel=1e-12, limit=400)[0]
        assert abs(orc.obtain_b(alpha) - want) <= 1e-9 * max(want, 1e-300) + 1e-300


@pytest.mark.parametrize("which", [0, 1])
def test_fam_burden_matches_numpy(which):
    N, K, U, S, X, y = make_family_case(30, 2, 23)
    rc, nul = orc.fastlmm_null(X, y, U, S)
    _, G, af = synth.make_gene(N, 10, seed=15, missing=0.02, common=True, mono=True, maf_hi=-0.7)
    rc, o = orc.fam_burden(G, X, y, U, S, nul, which)
    assert rc == 0 and o.fit_ok
    Gf, fl, kp = orc.flip_poly(G)
    n = (Gf.astype(int) > 0).sum(1)
    c = (n > 0).astype(float) if which == 0 else n.astype(float)
    beta = np.array(nul.beta[:2])
    lam = np.abs(S)
    Sinv = np.diag(1.0 / (lam + nul.delta))
    ux = U.T @ X
    ur = U.T @ y - ux @ beta
    ugc = U.T @ (c - c.mean())
    scaledK = Sinv - Sinv @ ux @ np.linalg.inv(ux.T @ Sinv @ ux) @ ux.T @ Sinv
    Us = np.sum(ugc * ur / (lam + nul.delta)) / nul.sigma2
    Vs = ugc @ scaledK @ ugc / nul.sigma2
    u1 = U.sum(0)
    afw = 0.5 * np.sum(u1 / lam * (U.T @ c)) / np.sum(u1 * u1 / lam)
    from scipy.stats import chi2
    assert o.num_site == Gf.shape[1]
    assert abs(o.U - Us) <= 1e-9 * abs(Us) and abs(o.V - Vs) <= 1e-9 * Vs
    assert abs(o.af - afw) <= 1e-10 * abs(afw)
    assert abs(o.pvalue - chi2.sf(Us * Us / Vs, 1)) <= 1e-9 * o.pvalue


def test_fam_score_of_a_raw_column_is_the_zeggini_score_of_a_01_column():
    """orc_fam_burden(which=2) (MetaScoreTest's MetaFamQtl: the raw column, unflipped) against which=1 on a column for
    which zegginiCollapse is the identity (values 0/1, allele frequency below one half), and GetNullCovB against
    numpy."""
    N, K, U, S, X, y = make_family_case(30, 2, 77)
    rc, nul = orc.fastlmm_null(X, y, U, S)
    assert rc == 0
    rng = np.random.default_rng(5)
    g = (rng.random(N) < 0.2).astype(float).reshape(-1, 1)
    rc1, a = orc.fam_burden(g, X, y, U, S, nul, 1)
    rc2, b = orc.fam_burden(g, X, y, U, S, nul, 2)
    assert rc1 == 0 and rc2 == 0
    assert b.U == pytest.approx(a.U, rel=1e-12) and b.V == pytest.approx(a.V, rel=1e-12)
    assert b.pvalue == pytest.approx(a.pvalue, rel=1e-12) and b.af == pytest.approx(a.af, rel=1e-12)
    # not flipped: g -> 2 - g changes the sign of U only
    rc3, c = orc.fam_burden(2.0 - g, X, y, U, S, nul, 2)
    assert rc3 == 0 and c.U == pytest.approx(-a.U, rel=1e-8) and c.V == pytest.approx(a.V, rel=1e-8)
    assert orc.fam_burden(np.ones((N, 1)), X, y, U, S, nul, 2)[0] == -1
    # which = 3 (MetaFamBinary: genotype not centred) against numpy: U = sum ug ur / (|S| + delta) / sigma2
    rc4, e = orc.fam_burden(g, X, y, U, S, nul, 3)
    beta = np.array([nul.beta[k] for k in range(X.shape[1])])
    ug, ur, sinv = U.T @ g[:, 0], U.T @ y - (U.T @ X) @ beta, 1.0 / (np.abs(S) + nul.delta)
    uxm = U.T @ X
    Kmat = np.diag(sinv) - (sinv[:, None] * uxm) @ np.linalg.inv(uxm.T @ (uxm * sinv[:, None])) @ (uxm * sinv[:, None]).T
    assert rc4 == 0
    assert e.U == pytest.approx((ug * ur * sinv).sum() / nul.sigma2, rel=1e-9)
    assert e.V == pytest.approx(ug @ Kmat @ ug / nul.sigma2, rel=1e-8)
    rc, covb = orc.fastlmm_covb(X, U, S, nul.delta)
    ux = U.T @ X
    want = np.linalg.inv(ux.T @ (ux * (np.abs(S) + nul.delta)[:, None]))
    assert rc == 0 and np.allclose(covb, want, rtol=1e-9)


def test_blockwise_numpy_statement_matches_literal_oracle():
    """The block-wise numpy statement of FamSKAT that the full-size GPU tests (tests/test_gpu_configs.py, N up to
    100 000) compare against equals the oracle's literal N x N restatement where that can still run."""
    import test_gpu_configs as t
    n_fam, d = 60, 3
    N, u4, s4, X, y, rng = t._family_blocks(n_fam, d, 5)
    U = t._dense_U(n_fam, u4).astype(np.float64)
    S = np.tile(s4, n_fam).astype(np.float64)
    assert np.abs((U * S) @ U.T - np.kron(np.eye(n_fam), t.BLK)).max() < 1e-6
    rc, nul = orc.fastlmm_null(X, y, U, S)
    assert rc == 0
    for M in (25, 3):
        G = t._gene_dropping(rng, n_fam, M)
        assert set(np.unique(G)) <= {0.0, 1.0, 2.0}
        rc, o = orc.famskat(G, X, y, U, S, nul)
        m, Q, ev = t._famskat_blockwise(G, X, y, u4, s4, nul.delta, nul.sigma2, np.array(nul.beta[:d]))
        assert rc == 0 and o.n_poly == m and o.n_lambda == len(ev)
        assert abs(o.Q - Q) <= 1e-10 * Q
        assert np.allclose(np.array(o.lambda_[:o.n_lambda]), ev, rtol=1e-9, atol=1e-11 * ev[0])
