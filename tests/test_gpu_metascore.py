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
