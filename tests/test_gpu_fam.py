"""GPU parity of the related-sample path through the C ABI: FastLMM null model (rvt_fit_fam_null) and FamSKAT
(rvt_run_fam_blocks) against the CPU oracle's literal N x N restatement."""
import numpy as np
import pytest

import orc
import synth
from test_fam_cpu import make_family_case

pytestmark = pytest.mark.gpu


@pytest.fixture
def eng():
    import rvtests_amd
    e = rvtests_amd.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("n_fam,d", [(40, 1), (60, 3)])
def test_fam_null_matches_oracle(eng, n_fam, d):
    N, K, U, S, X, y = make_family_case(n_fam, d, 3 + d)
    eng.set_kinship(U, S)
    nul = eng.fit_fam_null(X, y)
    rc, onul = orc.fastlmm_null(X, y, U, S)
    assert rc == 0
    assert nul.max_index == onul.max_index and nul.brent_evals >= 4
    # The likelihood is flat at its optimum (differences of 1e-12 between Brent's trial points), so rounding-level
    # differences in the function values can change a comparison inside Brent and with it the trajectory; what the
    # reference's stopping rule (bracket < 1e-3 ABSOLUTE, GSLMinimizer.cpp:8) pins down is delta to ~1e-3 and
    # beta / sigma2 to the corresponding accuracy.  When the trajectories coincide the numbers agree to 1e-8.
    assert abs(nul.delta - onul.delta) <= 2e-3 + 1e-3 * onul.delta
    if nul.brent_evals == onul.brent_evals:
        assert abs(nul.delta - onul.delta) <= 1e-7 * onul.delta
        assert abs(nul.sigma2_g - onul.sigma2) <= 1e-8 * onul.sigma2
        assert np.allclose(nul.beta[:d], onul.beta[:d], rtol=1e-8, atol=1e-10)
    assert abs(nul.sigma2_g - onul.sigma2) <= 5e-3 * onul.sigma2
    assert np.allclose(nul.beta[:d], onul.beta[:d], rtol=5e-3, atol=5e-4)


@pytest.mark.parametrize("n_fam,d,Ms", [(40, 2, (12, 1, 30)), (75, 3, (20, 45, 7))])
def test_famskat_matches_oracle(eng, n_fam, d, Ms):
    N, K, U, S, X, y = make_family_case(n_fam, d, 20 + d)
    eng.set_kinship(U, S)
    nul = eng.fit_fam_null(X, y)
    # FamSkat::FitNullModel takes (delta, sigma2, beta) from FastLMM: give the oracle the device's values so that this
    # test pins the FamSKAT stage itself (the null fit has its own test above)
    onul = orc.FamNull()
    onul.ok = 1
    onul.delta, onul.sigma2 = nul.delta, nul.sigma2_g
    for k in range(d):
        onul.beta[k] = nul.beta[k]
    genes = [synth.make_gene(N, M, seed=300 + M, missing=0.02, common=True, mono=(M > 5))[1] for M in Ms]
    genes.append(np.ones((N, 3)))                       # all monomorphic: NA row
    ptrs = [eng.upload_block(G) for G in genes]
    out = eng.run_fam_blocks(ptrs, [G.shape[1] for G in genes])
    for r, G in zip(out, genes):
        rc, o = orc.famskat(G, X, y, U, S, onul)
        assert r.n_variants == G.shape[1] and r.n_poly == o.n_poly
        if rc != 0:
            assert r.famskat_ok == 0
            continue
        assert r.famskat_ok == 1
        assert abs(r.famskat_Q - o.Q) <= 1e-7 * o.Q
        assert r.skat_nlambda == o.n_lambda
        assert abs(r.famskat_p - o.pvalue) <= 2e-6 * abs(o.pvalue) + 1e-12


@pytest.mark.parametrize("n_fam,d,V", [(40, 2, 25), (60, 3, 110)])
def test_metacov_fam_matches_oracle(eng, n_fam, d, V):
    """MetaCov with kinship (quantitative): rvt_cov_block_fam against the oracle's MetaCovFamQtl restatement."""
    N, K, U, S, X, y = make_family_case(n_fam, d, 50 + d)
    eng.set_kinship(U, S)
    nul = eng.fit_fam_null(X, y)
    onul = orc.FamNull()
    onul.ok = 1
    onul.delta, onul.sigma2 = nul.delta, nul.sigma2_g
    for k in range(d):
        onul.beta[k] = nul.beta[k]
    _, G, af = synth.make_gene(N, V, seed=600 + V, missing=0.02, common=True, mono=True)
    rng = np.random.default_rng(5)
    pos = np.cumsum(rng.integers(1, 300, V)).astype(np.int32)
    chrom = np.ones(V, dtype=np.int32)
    ptr = eng.upload_block(G)
    cov, xz, zz, poly = eng.cov_block_fam(ptr, V, d)
    rc, kept, ocov, row_end, oxz, ozz = orc.metacov_fam(G, chrom, pos, X, U, S, onul, 2500)
    assert rc == 0
    assert (poly == kept).all()
    m = ~np.isnan(ocov)
    assert m.sum() > V
    assert np.abs(cov[m] - ocov[m]).max() < 1e-8 * np.abs(ocov[m]).max()
    kk = kept.astype(bool)
    assert np.allclose(xz[kk], oxz[kk], rtol=1e-8, atol=1e-9 * max(np.abs(oxz[kk]).max(), 1.0))
    assert np.allclose(zz, ozz, rtol=1e-9)


def test_metacov_fam_binary_scale(eng):
    """MetaCovFamBinary: the family covariance of a 0/1 phenotype scaled by b^2 (b = obtainB(alpha))."""
    N, K, U, S, X, y = make_family_case(50, 2, 77)
    yb = (y > np.median(y)).astype(np.float64)
    yb[:7] = 1.0
    eng.set_kinship(U, S)
    nul = eng.fit_fam_null(X, yb)
    alpha, b = eng.fam_binary_scale(int(yb.sum()), int(N - yb.sum()))
    assert abs(b - orc.obtain_b(alpha)) <= 2e-7 * b              # float storage of b
    onul = orc.FamNull()
    onul.ok = 1
    onul.delta, onul.sigma2 = nul.delta, nul.sigma2_g
    for k in range(2):
        onul.beta[k] = nul.beta[k]
    V = 20
    _, G, af = synth.make_gene(N, V, seed=41, missing=0.02, common=True, mono=True)
    pos = np.cumsum(np.random.default_rng(6).integers(1, 300, V)).astype(np.int32)
    chrom = np.ones(V, dtype=np.int32)
    ptr = eng.upload_block(G)
    cov, xz, zz, poly = eng.cov_block_fam(ptr, V, 2)
    rc, kept, ocov, row_end, oxz, ozz = orc.metacov_fam_binary(G, chrom, pos, X, yb, U, S, onul, 2500)
    assert rc == 0 and (poly == kept).all()
    m = ~np.isnan(ocov)
    assert np.abs(cov[m] - ocov[m]).max() < 1e-6 * np.abs(ocov[m]).max()      # b is a float on both sides
    assert np.allclose(zz, ozz, rtol=1e-6)


@pytest.mark.parametrize("n_fam,d,V,col0,H,binary", [(40, 2, 60, 0, 60, 0), (60, 3, 110, 17, 40, 0),
                                                      (50, 2, 48, 5, 9, 1)])
def test_metacov_fam_rectangle_matches_block_and_oracle(eng, n_fam, d, V, col0, H, binary):
    """rvt_cov_rect_fam (windows wider than one block: integer-plane products of the rotated columns) against the block
    kernel on the same columns and against the oracle's MetaCovFamQtl / MetaCovFamBinary restatement."""
    N, K, U, S, X, y = make_family_case(n_fam, d, 150 + d)
    if binary:
        y = (y > np.median(y)).astype(np.float64)
    eng.set_kinship(U, S)
    nul = eng.fit_fam_null(X, y)
    if binary:
        eng.fam_binary_scale(int(y.sum()), int(N - y.sum()))
    onul = orc.FamNull()
    onul.ok = 1
    onul.delta, onul.sigma2 = nul.delta, nul.sigma2_g
    for k in range(d):
        onul.beta[k] = nul.beta[k]
    _, G, af = synth.make_gene(N, V, seed=900 + V, missing=0.02, common=True, mono=True)
    pos = np.cumsum(np.random.default_rng(8).integers(1, 300, V)).astype(np.int32)
    chrom = np.ones(V, dtype=np.int32)
    ptr = eng.upload_block(G)
    W = V - col0
    bcov, bxz, bzz, bpoly = eng.cov_block_fam(ptr, V, d)
    rcov, rxz, rzz, rpoly = eng.cov_rect_fam(ptr, col0, H, W, d)
    assert (rpoly == bpoly[col0:]).all()
    scale = np.abs(bcov[~np.isnan(bcov)]).max()
    for h in range(H):
        a, b = rcov[h, h:], bcov[col0 + h, col0 + h:]
        assert np.abs(a - b).max() <= 1e-9 * scale, h
    assert np.allclose(rxz, bxz[col0:], rtol=1e-8, atol=1e-9 * max(np.abs(bxz).max(), 1.0))
    assert np.allclose(rzz, bzz, rtol=1e-12)
    if binary:
        rc, kept, ocov, row_end, oxz, ozz = orc.metacov_fam_binary(G, chrom, pos, X, y, U, S, onul, 10 ** 7)
        tol = 1e-6
    else:
        rc, kept, ocov, row_end, oxz, ozz = orc.metacov_fam(G, chrom, pos, X, U, S, onul, 10 ** 7)
        tol = 1e-8
    assert rc == 0
    sub = ocov[col0:col0 + H, col0:]
    m = ~np.isnan(sub)
    assert m.sum() > H
    assert np.abs(rcov[m] - sub[m]).max() < tol * np.abs(sub[m]).max()


@pytest.mark.parametrize("n_fam,d", [(40, 2), (70, 3)])
def test_fam_burden_matches_oracle(eng, n_fam, d):
    """FamCMC / FamZeggini (collapse + FastLMM score test + GLS allele frequency) together with FamSKAT in one call."""
    N, K, U, S, X, y = make_family_case(n_fam, d, 90 + d)
    eng.set_kinship(U, S)
    nul = eng.fit_fam_null(X, y)
    onul = orc.FamNull()
    onul.ok = 1
    onul.delta, onul.sigma2 = nul.delta, nul.sigma2_g
    for k in range(d):
        onul.beta[k] = nul.beta[k]
    genes = [synth.make_gene(N, M, seed=800 + M, missing=0.02, common=True, mono=(M > 5), maf_hi=-0.8)[1]
             for M in (14, 2, 33, 9)]
    genes.append(np.zeros((N, 3)))                      # no polymorphic column
    ptrs = [eng.upload_block(G) for G in genes]
    out = eng.run_fam_blocks(ptrs, [G.shape[1] for G in genes], tests=16 | 32 | 64)
    for r, G in zip(out, genes):
        for which, ok, af, u, v, p in ((0, r.famcmc_ok, r.famcmc_af, r.famcmc_U, r.famcmc_V, r.famcmc_p),
                                       (1, r.famzeg_ok, r.famzeg_af, r.famzeg_U, r.famzeg_V, r.famzeg_p)):
            rc, o = orc.fam_burden(G, X, y, U, S, onul, which)
            if rc != 0:
                assert ok == 0
                continue
            assert ok == 1 and r.n_poly == o.num_site
            assert abs(u - o.U) <= 1e-8 * abs(o.U) + 1e-12
            assert abs(v - o.V) <= 1e-8 * o.V
            assert abs(af - o.af) <= 1e-9 * abs(o.af) + 1e-15
            assert abs(p - o.pvalue) <= 1e-6 * o.pvalue
        rc, s = orc.famskat(G, X, y, U, S, onul)
        assert r.famskat_ok == (1 if rc == 0 else 0)
        if rc == 0:
            assert abs(r.famskat_Q - s.Q) <= 1e-7 * s.Q


def test_rotation_k_range_cut_is_exact(monkeypatch):
    """The integer GEMM accumulates a K range in int32 and cuts ranges that could overflow (planes_gemm); with the cut
    forced to 128 samples the rotated statistics must not change by more than the fp64 re-association of the pieces."""
    import rvtests_amd
    N, K, U, S, X, y = make_family_case(60, 2, 11)
    genes = [synth.make_gene(N, M, seed=700 + M, missing=0.02, common=True)[1] for M in (9, 31)]
    outs = []
    for kmax in (None, "128"):
        if kmax:
            monkeypatch.setenv("RVT_ROT_KMAX", kmax)
        e = rvtests_amd.Engine(0)
        try:
            e.set_kinship(U, S)
            e.fit_fam_null(X, y)
            ptrs = [e.upload_block(G) for G in genes]
            outs.append([(r.famskat_Q, r.famskat_p) for r in e.run_fam_blocks(ptrs, [G.shape[1] for G in genes])])
        finally:
            e.close()
    for (q0, p0), (q1, p1) in zip(*outs):
        assert abs(q0 - q1) <= 1e-12 * abs(q0) and abs(p0 - p1) <= 1e-9 * p0 + 1e-14


def test_family_structure_is_detected_and_changes_nothing(monkeypatch):
    """Block-diagonal kinship (families of mixed size, eigenpairs handed over sorted by eigenvalue, i.e. scattered over
    the families): rvt_set_kinship clusters the eigenvectors by support and the rotation skips the zero blocks.  The
    results must be those of the dense product (RVT_KINSHIP_DENSE=1) to rounding, and the oracle's."""
    import rvtests_amd
    rng = np.random.default_rng(5)
    sizes = rng.choice([1, 2, 3, 4, 6], size=700)
    N = int(sizes.sum())
    U = np.zeros((N, N))
    S = np.zeros(N)
    o = 0
    for sz in sizes:                                   # a random positive-definite "kinship" per family
        A = rng.uniform(0.1, 0.5, (sz, sz))
        Kf = (A + A.T) / 2 + np.eye(sz)
        s, u = np.linalg.eigh(Kf)
        U[o:o + sz, o:o + sz] = u
        S[o:o + sz] = s
        o += sz
    order = np.argsort(S, kind="stable")               # what an eigen-solver hands over: ascending eigenvalues
    U, S = np.asfortranarray(U[:, order].astype(np.float32)), S[order].astype(np.float32)
    d = 3
    X = np.column_stack([np.ones(N)] + [rng.standard_normal(N) for _ in range(d - 1)])
    y = 0.3 * X[:, 1] + rng.standard_normal(N)
    genes = [synth.make_gene(N, M, seed=800 + M, missing=0.01, common=True)[1] for M in (5, 24, 40)]
    outs, nulls, fracs = [], [], []
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("RVT_KINSHIP_DENSE", "1")
        e = rvtests_amd.Engine(0)
        try:
            e.set_kinship(U, S)
            fracs.append(e.kinship_structure())
            nulls.append(e.fit_fam_null(X, y))
            ptrs = [e.upload_block(G) for G in genes]
            outs.append([(r.famskat_ok, r.n_poly, r.famskat_Q, r.famskat_p)
                         for r in e.run_fam_blocks(ptrs, [G.shape[1] for G in genes])])
        finally:
            e.close()
    assert fracs[0] < 0.25 and fracs[1] == 1.0
    assert nulls[0].brent_evals == nulls[1].brent_evals and abs(nulls[0].delta - nulls[1].delta) <= 1e-9 * nulls[1].delta
    for (ok0, n0, q0, p0), (ok1, n1, q1, p1) in zip(*outs):
        assert ok0 == ok1 == 1 and n0 == n1
        assert abs(q0 - q1) <= 1e-11 * abs(q1) and abs(p0 - p1) <= 1e-8 * p1 + 1e-14
    rc, onul = orc.fastlmm_null(X, y, U.astype(np.float64), S.astype(np.float64))
    assert rc == 0
    for (ok0, n0, q0, p0), G in zip(outs[0], genes):
        rc, s = orc.famskat(G, X, y, U.astype(np.float64), S.astype(np.float64), onul)
        assert rc == 0 and abs(q0 - s.Q) <= 1e-6 * s.Q
