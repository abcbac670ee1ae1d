"""MetaScore oracle (oracle/orc_models.cpp: orc_metascore) against an independent numpy / scipy statement of the
reference's formulas (src/Model.h:3516-3549 quantitative, :3706-3769 binary; LinearRegressionScoreTest.cpp:173-263,
LogisticRegressionScoreTest.cpp:220-302).  The reference ships no golden MetaScore output and its score tests need
Eigen + GSL (not buildable here), so this restatement is pinned by algebra only — DESIGN.md says so."""
import numpy as np
import pytest
from scipy import stats

import orc
from test_metacov_cpu import make_case


def numpy_metascore(G, X, y, binary):
    N, V = G.shape
    if not binary:
        beta = np.linalg.solve(X.T @ X, X.T @ y)
        res = y - X @ beta
        s2 = res @ res / N
        w = np.ones(N)
        covb = np.diag(np.linalg.inv(X.T @ X)) * s2
    else:
        rc, beta, p, w = orc.fit_logistic(X, y)
        assert rc == 0
        res, s2 = y - p, 1.0
        covb = np.diag(np.linalg.inv(X.T @ (X * w[:, None])))
    out = dict(ok=np.zeros(V, int), U=np.zeros(V), V=np.zeros(V), effect=np.zeros(V), se=np.zeros(V), p=np.ones(V))
    ZZi = np.linalg.inv(X.T @ (X * w[:, None]))
    for h in range(V):
        g = G[:, h]
        if len(np.unique(g)) == 1:
            continue
        U = g @ res
        sz = g @ (X * w[:, None])
        SS = g @ (g * w) - sz @ ZZi @ sz
        if not binary:
            out["U"][h], out["V"][h] = U / s2, SS / s2
            out["effect"][h], out["se"][h] = U / SS, np.sqrt(s2 / SS)
            stat = U * U / (SS * s2)
        else:
            out["U"][h], out["V"][h] = U, SS
            out["effect"][h], out["se"][h] = U / SS, 1 / np.sqrt(SS)
            stat = U * U / SS
        out["p"][h] = stats.chi2.sf(stat, 1)
        out["ok"][h] = 1
    return out, beta, covb, s2


@pytest.mark.parametrize("binary", [0, 1])
@pytest.mark.parametrize("V,d", [(9, 1), (33, 3), (70, 5)])
def test_oracle_matches_numpy(binary, V, d):
    G, chrom, pos, X, y = make_case(900, V, d, binary, 500 + V + d + binary)
    G[:, 2] = 1.0                                 # a monomorphic site
    G[:, 5] = np.where(G[:, 5] > 0, 0.37, 0.0)    # imputed dosages are not integers
    rc, o = orc.metascore(G, X, y, binary)
    assert rc == 0
    ref, beta, covb, s2 = numpy_metascore(G, X, y, binary)
    assert (o["ok"] == ref["ok"]).all() and o["ok"][2] == 0 and o["ok"].sum() >= 3
    for k in ("U", "V", "effect", "se", "p"):
        assert np.allclose(o[k], ref[k], rtol=1e-9, atol=1e-300), k
    assert np.allclose(o["beta"], beta, rtol=1e-8, atol=1e-10)
    assert np.allclose(o["covb"], covb, rtol=1e-8)
    assert o["sigma2"] == pytest.approx(s2, rel=1e-10)


def test_no_flip():
    """MetaScore does not flip a column with AF > 0.5: recoding g -> 2 - g changes the sign of U and of the effect and
    leaves V and the p-value alone."""
    G, chrom, pos, X, y = make_case(700, 12, 2, 0, 321)
    rc, o = orc.metascore(G, X, y, 0)
    assert rc == 0
    rc, of = orc.metascore(2.0 - G, X, y, 0)
    assert rc == 0
    k = o["ok"].astype(bool)
    assert k.sum() >= 3 and (of["ok"] == o["ok"]).all()
    assert np.allclose(of["U"][k], -o["U"][k], rtol=1e-8, atol=1e-9)
    assert np.allclose(of["V"][k], o["V"][k], rtol=1e-8)
    assert np.allclose(of["p"][k], o["p"][k], rtol=1e-7)
    assert np.allclose(of["effect"][k], -o["effect"][k], rtol=1e-8, atol=1e-12)
