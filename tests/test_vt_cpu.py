"""CPU: AnalyticVT.  (1) The band probability: the device's deterministic lattice rule (rvt_mvn.h compiled for the host)
and the oracle's independent Halton rule against the REFERENCE's own MVTDST (regression/libMvtnorm/mvt.f compiled with
flang, oracle/_ref/libref_mvt.so) — to within the accuracy that randomised rule has (abseps 1e-3), and against each
other far tighter.  (2) The oracle's literal restatement of AnalyticVT::fit against a direct numpy statement."""
import numpy as np
import pytest

import hc
import orc
import synth


def _nested_cor(rng, K, base=40):
    """Correlation of nested partial sums of correlated scores — the structure v_phi has."""
    m = K + 5
    A = rng.standard_normal((base, m))
    V = A.T @ A / base + 0.2 * np.eye(m)
    idx = np.sort(rng.choice(np.arange(1, m + 1), K, replace=False))
    Phi = np.zeros((m, K))
    for j, c in enumerate(idx):
        Phi[:c, j] = 1.0
    Vp = Phi.T @ V @ Phi
    s = np.sqrt(np.diag(Vp))
    return Vp / np.outer(s, s)


def test_phiinv_round_trip():
    from math import erfc, sqrt
    for p in (1e-300, 1e-12, 1e-4, 0.01, 0.3, 0.5, 0.7, 0.999, 1 - 1e-12):
        x = hc.mvn_phiinv(p)
        assert abs(0.5 * erfc(-x / sqrt(2)) - p) <= 1e-14 * max(p, 1e-300) + 1e-16


@pytest.mark.parametrize("K,T", [(2, 1.0), (3, 2.2), (8, 2.5), (25, 3.0), (60, 2.0), (60, 4.0)])
def test_band_probability_three_ways(K, T):
    rng = np.random.default_rng(100 + K)
    R = _nested_cor(rng, K)
    p_dev, e_dev = hc.mvn_band(R, T)
    p_orc, e_orc = orc.mvn_band(R, T, points=2048)
    assert abs(p_dev - p_orc) <= e_dev + e_orc + 1e-5      # two independent deterministic rules
    assert e_dev < (5e-4 if K <= 25 else 2e-3)
    if orc.ref_mvt() is None:
        pytest.skip("reference MVTDST not built on this machine")
    for seed in (1, 7):
        inform, p_ref, e_ref = orc.ref_mvn_band(R, T, seed)
        # inform = 1: the reference's own error estimate exceeds abseps = 1e-3 — MvtNorm::compute_Band then returns -1
        # and AnalyticVT prints NA (high-dimensional cases with a moderate threshold)
        assert inform in (0, 1) and (inform == 1) == (e_ref > 1e-3)
        assert abs(p_dev - p_ref) <= e_ref + e_dev        # within the reference's own error estimate
        assert abs(p_orc - p_ref) <= e_ref + e_orc


def test_band_probability_simple_cases():
    from math import erf, sqrt
    for T in (0.5, 1.96, 3.3):
        one = erf(T / sqrt(2))
        p, e = hc.mvn_band(np.eye(1), T)
        assert abs(p - one) < 1e-15
        p, e = hc.mvn_band(np.eye(4), T)                     # independent: product of the margins
        assert abs(p - one ** 4) < 1e-9
        R = np.ones((3, 3))                                   # perfectly correlated: singular, two zero pivots
        p, e = hc.mvn_band(R, T)
        assert abs(p - one) < 1e-9


@pytest.mark.parametrize("N,M,d,seed", [(300, 12, 1, 1), (500, 30, 3, 2), (400, 7, 2, 3)])
def test_oracle_matches_numpy_statement(N, M, d, seed):
    af, G, af2 = synth.make_gene(N, M, seed=seed, missing=0.02, common=True, mono=True)
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=seed + 10)
    rc, o, cor = orc.analytic_vt(G, af2, X, y, mvn_points=512)
    assert rc == 0 and o.fit_ok
    # numpy: residualise, u, v, thresholds
    Gf = G.copy()
    s = Gf.sum(0)
    Gf[:, s > N] = 2 - Gf[:, s > N]
    keep = np.array([len(np.unique(Gf[:, j])) > 1 for j in range(M)])
    Gk = Gf[:, keep]
    H = X @ np.linalg.solve(X.T @ X, X.T)
    Rm = np.eye(N) - H
    x = Rm @ Gk
    x -= x.mean(0)
    yr = Rm @ y
    sig = np.var(yr)
    u = x.T @ yr
    V = x.T @ x * sig
    m = Gk.shape[1]
    maf = np.minimum(af2[:m], 1 - af2[:m])
    ok = (maf >= 1e-10) & (np.diag(V) >= 1e-10)
    keys = sorted({int(np.ceil(f * 1e6)) for f, k in zip(maf, ok) if k})
    cut = np.array(keys) / 1e6
    Phi = ((maf[:, None] <= cut[None, :]) & ok[:, None]).astype(float)
    up, vp = u @ Phi, Phi.T @ V @ Phi
    t = np.abs(up / np.sqrt(np.diag(vp)))
    j = int(np.argmax(t))
    assert o.n_cutoff == len(cut) and o.opt_num == int(Phi[:, j].sum())
    assert abs(o.stat - t[j]) <= 1e-9 * t[j] and abs(o.U - up[j]) <= 1e-9 * abs(up[j]) + 1e-12
    assert abs(o.V - vp[j, j]) <= 1e-9 * vp[j, j] and o.opt_maf == cut[j]
    assert o.min_maf == maf.min() and o.max_maf == maf.max()
    sd = np.sqrt(np.diag(vp))
    assert np.allclose(cor, vp / np.outer(sd, sd), rtol=1e-9, atol=1e-12)
    p_dev, e_dev = hc.mvn_band(cor, o.stat)
    assert abs((1 - p_dev) - o.pvalue) <= 5e-4 + o.p_err + e_dev
