"""MetaCov oracle (oracle/orc_models.cpp: orc_metacov) against an independent numpy statement of the reference's
formulas (src/Model.cpp:506-593 quantitative, :694-778 binary, window rule src/Model.h:3956-3990)."""
import numpy as np
import pytest

import orc
import synth


def numpy_metacov(G, chrom, pos, X, y, binary, window):
    N, V = G.shape
    d = X.shape[1]
    kept = np.array([len(np.unique(G[:, j])) > 1 for j in range(V)])
    if not binary:
        beta = np.linalg.solve(X.T @ X, X.T @ y)
        res = y - X @ beta
        s2 = res @ res / N
        Gc = G - G.mean(0)
        XX = Gc.T @ Gc / s2
        XZ = Gc.T @ X / s2
        Xc = X - X.mean(0)
        ZZ = Xc.T @ Xc / s2
        ZZi = np.zeros((d, d))
        if d > 1:
            ZZi[1:, 1:] = np.linalg.inv(ZZ[1:, 1:])
    else:
        rc, beta, p, v = orc.fit_logistic(X, y)
        assert rc == 0
        XX = G.T @ (G * v[:, None])
        XZ = G.T @ (X * v[:, None])
        ZZ = X.T @ (X * v[:, None])
        ZZi = np.linalg.inv(ZZ)
    val = XX - XZ @ ZZi @ XZ.T
    cov = np.full((V, V), np.nan)
    row_end = np.full(V, -1)
    for h in range(V):
        if not kept[h]:
            continue
        for j in range(h, V):
            if chrom[j] != chrom[h] or abs(pos[j] - pos[h]) > window:
                break
            if kept[j]:
                cov[h, j] = val[h, j]
                row_end[h] = j
    return kept.astype(int), cov, row_end, XZ, ZZ


def make_case(N, V, d, binary, seed):
    rng = np.random.default_rng(seed)
    _, G, af = synth.make_gene(N, V, seed=seed, missing=0.01, common=True, mono=True)
    X, y, res, v, s2 = synth.make_null(N, d, binary, seed=seed + 1)
    chrom = np.ones(V, dtype=np.int32)
    chrom[V * 2 // 3:] = 2
    pos = np.cumsum(rng.integers(1, 400, V)).astype(np.int32)
    return G, chrom, pos, X, y


@pytest.mark.parametrize("binary", [0, 1])
@pytest.mark.parametrize("d", [1, 3])
def test_oracle_matches_numpy(binary, d):
    G, chrom, pos, X, y = make_case(400, 23, d, binary, 11 + d + 7 * binary)
    rc, kept, cov, row_end, xz, zz = orc.metacov(G, chrom, pos, X, y, binary, 1500)
    assert rc == 0
    k2, c2, r2, xz2, zz2 = numpy_metacov(G, chrom, pos, X, y, binary, 1500)
    assert (kept == k2).all() and (row_end == r2).all()
    assert (np.isnan(cov) == np.isnan(c2)).all()
    m = ~np.isnan(c2)
    scale = np.abs(c2[m]).max()
    assert np.abs(cov[m] - c2[m]).max() < 1e-9 * scale
    kk = kept.astype(bool)
    # (for d == 1 the quantitative covXZ is the centred genotype against the intercept: zero up to rounding)
    assert np.allclose(xz[kk], xz2[kk], rtol=1e-9, atol=1e-9 * max(np.abs(xz2).max(), 1.0))
    assert np.allclose(zz, zz2, rtol=1e-9, atol=1e-9 * np.abs(zz2).max())


def test_oracle_float_path_is_close():
    """use_float=1 restates the reference's fp32 storage: same rows, values within fp32 noise of the fp64 ones."""
    G, chrom, pos, X, y = make_case(400, 17, 3, 0, 5)
    rc, kept, cov, row_end, xz, zz = orc.metacov(G, chrom, pos, X, y, 0, 2000)
    rc2, kept2, covf, row_end2, xzf, zzf = orc.metacov(G, chrom, pos, X, y, 0, 2000, use_float=True)
    assert rc == 0 and rc2 == 0 and (row_end == row_end2).all()
    m = ~np.isnan(cov)
    assert np.abs(covf[m] - cov[m]).max() < 2e-4 * np.abs(cov[m]).max()
