"""Seeded synthetic genes for the tests (shape of BASELINE.json configs 2-4, scaled down)."""
import numpy as np

import orc


def make_gene(N, M, seed, missing=0.0, common=False, mono=False, maf_lo=-3.3, maf_hi=-1.3):
    """Returns (Graw with -9 for missing, G imputed as dc.consolidate leaves it, counter AF)."""
    rng = np.random.default_rng(seed)
    maf = 10 ** rng.uniform(maf_lo, maf_hi, M)
    Graw = rng.binomial(2, maf, size=(N, M)).astype(np.float64)
    if missing > 0:
        Graw[rng.random((N, M)) < missing] = -9.0
    if common and M > 3:
        Graw[:, 2] = np.where(Graw[:, 2] < 0, -9.0, 2.0 - Graw[:, 2])  # alt allele common -> flipped
    if mono and M > 5:
        Graw[:, 4] = 0.0                                                 # monomorphic reference
    if mono and M > 7:
        Graw[:, 1] = np.where(Graw[:, 1] < 0, -9.0, 2.0)                 # monomorphic after the flip
    af = orc.counter_af(Graw)
    G = orc.impute_mean(Graw)
    return Graw, G, af


def make_null(N, d, binary, seed, G_effect=None):
    """Covariates, phenotype and the fitted null model (through the oracle)."""
    rng = np.random.default_rng(seed)
    X = np.column_stack([np.ones(N)] + [rng.normal(size=N) for _ in range(d - 1)])
    lin = np.zeros(N)
    if d > 1:
        lin += 0.3 * X[:, 1]
    if d > 2:
        lin -= 0.2 * X[:, 2]
    if G_effect is not None:
        lin += G_effect
    if binary:
        y = (rng.random(N) < 1.0 / (1.0 + np.exp(-(-1.0 + lin)))).astype(np.float64)
        rc, beta, p, v = orc.fit_logistic(X, y)
        assert rc == 0
        return X, y, y - p, v, 1.0
    y = lin + rng.normal(size=N)
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    assert rc == 0
    return X, y, res, np.full(N, s2), s2
