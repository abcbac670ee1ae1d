"""Device null-model fitters (rvt_fit_null) against the oracle's restatement of LinearRegression::FitLinearModel and
LogisticRegression::FitLogisticModel, and end to end: a gene tested under the device-fitted null gives the same
statistics as under the oracle-fitted null handed to rvt_set_null."""
import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def eng():
    import rvtests_amd
    e = rvtests_amd.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("binary,d,N", [(0, 1, 700), (0, 4, 5000), (1, 1, 900), (1, 3, 6000)])
def test_fit_null_matches_oracle(eng, binary, d, N):
    import rvtests_amd
    _, G, af = synth.make_gene(N, 18, seed=3 + d, missing=0.01, common=True, mono=True)
    X, y, res, v, s2 = synth.make_null(N, d, binary, seed=5 + d, G_effect=0.4 * G[:, :3].sum(1))
    beta, sigma2 = eng.fit_null(binary, X, y)
    if binary:
        rc, obeta, p, ov = orc.fit_logistic(X, y)
        assert rc == 0
        assert np.allclose(beta, obeta, rtol=1e-9, atol=1e-11)
    else:
        rc, obeta, pred, ores, os2 = orc.fit_linear(X, y)
        assert rc == 0
        assert np.allclose(beta, obeta, rtol=1e-9, atol=1e-11)
        assert abs(sigma2 - os2) <= 1e-11 * os2
    ptr = eng.upload_block(G)
    a = eng.run_blocks([ptr], [G.shape[1]], [af])[0]
    eng.set_null(binary, X, res, v, s2)          # synth.make_null's own (oracle-style) fit
    ptr2 = eng.upload_block(G)
    b = eng.run_blocks([ptr2], [G.shape[1]], [af])[0]
    for f in ("skat_Q", "skat_p", "skato_Q", "skato_p", "cmc_p", "zeg_p"):
        x, yv = getattr(a, f), getattr(b, f)
        assert abs(x - yv) <= 1e-7 * abs(yv) + 1e-13, f
    assert a.cmc_nonref == b.cmc_nonref and a.skato_rho == b.skato_rho


def test_fit_null_reports_failure(eng):
    N = 300
    X = np.column_stack([np.ones(N), np.ones(N)])          # collinear: X'X singular
    y = np.random.default_rng(1).standard_normal(N)
    import rvtests_amd
    with pytest.raises(rvtests_amd.RvtError):
        eng.fit_null(0, X, y)
