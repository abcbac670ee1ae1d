"""BASELINE.json configs[0] / SURVEY §8d config 1: the reference's own example (example/example.vcf, example/pheno,
example/setFile: N = 9 phenotyped samples, set1 = 3 variants, trait y1, no covariates, `--kernel skat` with its default
nPerm = 10000, alpha = 0.05, beta1 = 1, beta2 = 25).  The fixture tests/golden/example_config1.json holds the data as
rvtests reads it; the reference ships no expected output for the command, so the numbers are the oracle's, checked here
against a direct numpy evaluation and on the GPU through the C ABI."""
import json
import os

import numpy as np
import pytest
from scipy import stats

import orc

HERE = os.path.dirname(os.path.abspath(__file__))


def load_case(trait="y1"):
    fx = json.load(open(os.path.join(HERE, "golden", "example_config1.json")))
    G = np.asfortranarray(np.array(fx["genotype_by_variant"], dtype=np.float64).T)      # N x M
    assert G.shape == (9, 3) and (G >= 0).all()                                          # no missing call in the set
    af = 0.5 * G.sum(0) / G.shape[0]                                                     # GenotypeCounter::getAF
    if trait == "y1":
        y = np.array(fx["y1"])
    else:
        y = np.array(fx["y4"], dtype=np.float64) - 1.0                                   # PLINK 1/2 -> 0/1
    X = np.ones((9, 1))
    return fx, G, af, X, y


def test_fixture_is_the_reference_example():
    fx, G, af, X, y = load_case()
    assert fx["samples"] == ["P%d" % i for i in range(1, 10)] and fx["sites"] == ["1:1", "1:2", "1:3"]
    assert G[:, 0].tolist() == [1, 0, 0, 2, 1, 0, 1, 1, 0]
    assert np.allclose(af, [6 / 18, 2 / 18, 1 / 18])


def test_oracle_on_config1_matches_numpy():
    fx, G, af, X, y = load_case()
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    assert rc == 0 and beta[0] == pytest.approx(y.mean()) and s2 == pytest.approx(((y - y.mean()) ** 2).mean())
    v = np.full(9, s2)
    rc, a = orc.skat(G, af, X, res, v, 0)
    assert rc == 0 and a.n_poly == 3
    # SkatTest weights (src/Model.h:2644-2661): beta_pdf(maf; 1, 25)^2;  Q = sum_j w_j (g_j' r)^2 (Skat.cpp:42-52)
    w = stats.beta.pdf(np.minimum(af, 1 - af), 1, 25) ** 2
    Q = float((w * (G.T @ res) ** 2).sum())
    assert a.Q == pytest.approx(Q, rel=1e-10)
    # eigenvalues of K^1/2 P0 K^1/2 with P0 = V - V 1 (1'V1)^-1 1'V (intercept only)
    P0 = s2 * (np.eye(9) - np.ones((9, 9)) / 9)
    K = np.sqrt(w)[:, None] * G.T
    lam = np.sort(np.linalg.eigvalsh(K @ P0 @ K.T))[::-1]
    got = np.array([a.lambda_[k] for k in range(a.n_lambda)]) if hasattr(a, "lambda_") else None
    if got is not None:
        assert np.allclose(got, lam[: len(got)], rtol=1e-9)
    rc, lit = orc.skat_literal(G, af, X, res, v, 0)
    assert rc == 0 and lit.pvalue == pytest.approx(a.pvalue, rel=1e-9) and 0 < a.pvalue < 1


@pytest.mark.gpu
@pytest.mark.parametrize("trait", ["y1", "y4"])
def test_gpu_on_config1_matches_oracle(trait):
    """`--kernel skat` (analytic + the default adaptive permutation test) plus SKAT-O / CMC / Zeggini on the example."""
    import rvtests_amd
    from test_gpu_parity import _check_gene
    fx, G, af, X, y = load_case(trait)
    binary = 0 if trait == "y1" else 1
    eng = rvtests_amd.Engine(0)
    try:
        eng.fit_null(binary, X, y)
        if binary:
            rc, beta, p, v = orc.fit_logistic(X, y)
            res = y - p
        else:
            rc, beta, pred, res, s2 = orc.fit_linear(X, y)
            v = np.full(9, s2)
        assert rc == 0
        prm = rvtests_amd.Params(1.0, 25.0, 1.0, 25.0, 10000, 0.05)
        ptr = eng.upload_block(G)
        eng.set_perm_exact(True)       # the reference's own rand() stream
        eng.rand_seed(1)
        r = eng.run_blocks([ptr], [3], [af], tests=rvtests_amd.TEST_ALL, params=prm)[0]
        _check_gene(r, G, af, X, y, res, v, binary, 1)
        orc.rand_seed(1)
        rc, pm = orc.skat_permute(G, af, res, r.skat_Q, 10000, 0.05)
        assert rc == 0 and r.perm_ok == 1
        assert (r.perm_actual_perm, r.perm_num_greater, r.perm_num_equal) == (pm.actual_perm, pm.num_x, pm.num_equal)
        assert r.perm_pvalue == pm.pvalue
    finally:
        eng.close()
