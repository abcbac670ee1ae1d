"""GPU: AnalyticVT through the C ABI (test bit RVT_TEST_ANALYTICVT) against the oracle's literal restatement: thresholds,
counts and statistics to rounding, the p-value to within the two deterministic rules' error estimates — and, where the
reference's own MVTDST is built, within ITS error estimate (the reference is a randomised rule at abseps 1e-3)."""
import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu

TEST_VT = 128


@pytest.fixture
def eng():
    import rvtests_amd
    e = rvtests_amd.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("N,d,Ms,extra", [(600, 1, (5, 18, 40), 0), (900, 3, (12, 33, 64, 1), 15)])
def test_analytic_vt_matches_oracle(eng, N, d, Ms, extra):
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=5 + d)
    eng.fit_null(0, X, y)
    genes = [synth.make_gene(N, M, seed=40 + M, missing=0.02, common=(M % 2 == 0), mono=(M > 6)) for M in Ms]
    ptrs = [eng.upload_block(G) for af0, G, af in genes]
    out = eng.run_blocks(ptrs, [g[1].shape[1] for g in genes], [g[2] for g in genes], tests=TEST_VT | extra)
    for r, (af0, G, af) in zip(out, genes):
        rc, o, cor = orc.analytic_vt(G, af, X, y, mvn_points=1024)
        if rc != 0:
            assert r.vt_ok == 0
            continue
        assert r.vt_ok == 1 and r.n_poly == o.n_poly
        assert r.vt_ncutoff == o.n_cutoff and r.vt_optnum == o.opt_num
        assert r.vt_minmaf == o.min_maf and r.vt_maxmaf == o.max_maf and r.vt_optmaf == o.opt_maf
        assert abs(r.vt_U - o.U) <= 1e-9 * abs(o.U) + 1e-11
        assert abs(r.vt_V - o.V) <= 1e-9 * o.V
        assert abs(r.vt_stat - o.stat) <= 1e-9 * o.stat
        assert abs(r.vt_p - o.pvalue) <= r.vt_p_error + o.p_err + 1e-5
        assert r.vt_p_error < 2e-3
        if orc.ref_mvt() is not None and o.n_cutoff > 1:
            inform, p_ref, e_ref = orc.ref_mvn_band(cor, o.stat, seed=3)
            assert abs((1.0 - p_ref) - r.vt_p) <= e_ref + r.vt_p_error
        if extra:                                   # the other tests of the same batch are not disturbed
            rc2, a = orc.skat(G, af, X, res, v, 0)
            if a.n_poly:
                assert abs(r.skat_p - a.pvalue) <= 1e-6 * a.pvalue + 1e-14


def test_analytic_vt_binary_trait_is_not_fitted(eng):
    """'Analytic VT test does not support binary outcomes. Results will be all NAs.' (src/Model.h:2143-2149)"""
    N = 500
    X, y, res, v, s2 = synth.make_null(N, 2, 1, seed=9)
    eng.fit_null(1, X, y)
    af0, G, af = synth.make_gene(N, 9, seed=3, missing=0.01, common=True)
    out = eng.run_blocks([eng.upload_block(G)], [9], [af], tests=TEST_VT)
    assert out[0].vt_ok == 0


@pytest.mark.parametrize("n_fam,d,Ms", [(40, 2, (9, 24)), (75, 3, (35, 6))])
def test_fam_analytic_vt_matches_oracle(eng, n_fam, d, Ms):
    """FamAnalyticVT (rvt_fam_analytic_vt): frequencies, scores and variances from the family-covariance machinery of
    the raw block with flips applied algebraically, against the oracle's literal N x N restatement."""
    from test_fam_cpu import make_family_case
    N, K, U, S, X, y = make_family_case(n_fam, d, 300 + d)
    eng.set_kinship(U, S)
    nul = eng.fit_fam_null(X, y)
    onul = orc.FamNull()
    onul.ok = 1
    onul.delta, onul.sigma2 = nul.delta, nul.sigma2_g
    for k in range(d):
        onul.beta[k] = nul.beta[k]
    genes = [synth.make_gene(N, M, seed=800 + M, missing=0.02, common=True, mono=(M > 7))[1] for M in Ms]
    genes[0][:, 1] = 2.0 - genes[0][:, 1] * (genes[0][:, 1] <= 2)          # a column that has to be flipped
    ptrs = [eng.upload_block(G) for G in genes]
    out = eng.fam_analytic_vt(ptrs, [G.shape[1] for G in genes])
    for r, G in zip(out, genes):
        rc, o, cor = orc.fam_analytic_vt(G, X, y, U, S, onul, mvn_points=1024)
        if rc != 0:
            assert r.vt_ok == 0
            continue
        assert r.vt_ok == 1 and r.n_poly == o.n_poly
        assert r.vt_ncutoff == o.n_cutoff and r.vt_optnum == o.opt_num
        assert abs(r.vt_optmaf - o.opt_maf) <= 1e-12 and abs(r.vt_minmaf - o.min_maf) <= 1e-9
        assert abs(r.vt_U - o.U) <= 1e-7 * abs(o.U) + 1e-10
        assert abs(r.vt_V - o.V) <= 1e-7 * o.V
        assert abs(r.vt_stat - o.stat) <= 1e-7 * o.stat
        assert abs(r.vt_p - o.pvalue) <= r.vt_p_error + o.p_err + 1e-5
