"""CPU: is the SKAT-O gap below p ~ 1e-7 rounding or logic?  (VERDICT round 3, weak #1 / next #4.)

The reference computes the SKAT-O p-value as `1 - integral` (regression/SkatO.cpp:256), so below ~1e-7 its digits are
absolute-accurate only: the integral is a double near 1 whose last bits come from ~10^3 Davies evaluations and a QAGS rule
with a 1.2e-4 relative tolerance.  This test measures, on causal genes whose p-values spread over the decades 1e-2 .. 1e-13:

  * the DEVICE algorithms compiled for the host (tests/hc.py: flip algebra, shared tridiagonal form, Sturm bisection,
    product-form Davies, QAGS state machine) against the oracle (Jacobi eigenvalues, term-by-term Davies, GSL-shaped QAGS);
  * the same with the device's Davies in its TERM-BY-TERM form (RVT_TEST_EXACT_DAVIES: bit-identical to the compiled
    reference's qf());
  * the ORACLE against ITSELF on a mathematically equivalent input (residuals and variance rescaled by c and c^2 with
    c = 1 + 2^-30: every product rounds differently, the exact p-value does not change).

If the three differences are of the same size — a few 1e-14 ABSOLUTE whatever the decade — the gap is the conditioning of
`1 - integral`, not a difference of logic; a logic difference would show as a RELATIVE difference that does not scale with
1 / p.  The table is written to gpurun_out/skato_smallp_cpu.json (copied to profiles/ by the collection script)."""
import json
import os

import numpy as np

import hc
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXACT_DAVIES = 1 << 31


def _genes(N, d, n_genes, seed):
    rng = np.random.default_rng(seed)
    X = np.column_stack([np.ones(N)] + [rng.normal(size=N) for _ in range(d - 1)])
    genes, eff = [], np.zeros(N)
    for k in range(n_genes):
        M = (18, 30, 44)[k % 3]
        maf = 10 ** rng.uniform(-2.6, -1.3, M)
        G = np.asfortranarray(rng.binomial(2, maf, size=(N, M)).astype(np.float64))
        burden = G[:, :5].sum(1)
        ncp = 8.0 + 9.0 * k                                  # graded non-centrality: p from 1e-2 down to ~1e-14
        eff += np.sqrt(ncp / (burden.var() * N)) * (burden - burden.mean())
        genes.append((G, G.sum(0) / (2.0 * N)))
    y = 0.3 * X[:, 1] + eff + rng.normal(size=N)
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    assert rc == 0
    return X, y, res, s2, genes


def test_skato_small_p_gap_is_conditioning_not_logic():
    N, d = 4000, 2
    X, y, res, s2, genes = _genes(N, d, 14, seed=41)
    v = np.full(N, s2)
    c = 1.0 + 2.0 ** -30
    rows = []
    for G, af in genes:
        rc, o = orc.skato(G, af, X, res, v, 0)
        assert rc == 0 and o.fit_ok
        rc2, o2 = orc.skato(G, af, X, res * c, v * (c * c), 0)         # the same p-value, every product rounded otherwise
        assert rc2 == 0
        h_fast, _, _, _ = hc.gene(G, af, X, res, v, 0, s2, tests=2)
        h_exact, _, _, _ = hc.gene(G, af, X, res, v, 0, s2, tests=2 | EXACT_DAVIES)
        assert h_fast.skato_ok and h_exact.skato_ok and h_fast.skato_rho == o.rho
        rows.append(dict(p=o.pvalue, M=G.shape[1], fast=abs(h_fast.skato_p - o.pvalue), exact=abs(h_exact.skato_p - o.pvalue),
                         self=abs(o2.pvalue - o.pvalue)))
    ps = np.array([r["p"] for r in rows])
    assert ps.min() < 1e-9 and ps.max() > 1e-3, "the planted effects no longer span the decades: %s" % ps
    table = {}
    for r in rows:
        key = "1e%d" % int(np.floor(np.log10(r["p"])))
        e = table.setdefault(key, dict(n=0, p_min=1.0, device_fast_abs=0.0, device_exact_davies_abs=0.0, oracle_self_abs=0.0))
        e["n"] += 1
        e["p_min"] = min(e["p_min"], r["p"])
        e["device_fast_abs"] = max(e["device_fast_abs"], r["fast"])
        e["device_exact_davies_abs"] = max(e["device_exact_davies_abs"], r["exact"])
        e["oracle_self_abs"] = max(e["oracle_self_abs"], r["self"])
    for e in table.values():
        for k in ("device_fast", "device_exact_davies", "oracle_self"):
            e[k + "_rel"] = e[k + "_abs"] / e["p_min"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "skato_smallp_cpu.json"), "w") as f:
        json.dump(dict(N=N, d=d, genes=len(genes), c="1 + 2^-30", by_decade=dict(sorted(table.items(), key=lambda kv: -int(kv[0][2:])))),
                  f, indent=1)
    # 1. where the reference's own digits are relative-accurate (p >= 1e-7) the device agrees to north_star's 1e-6
    for r in rows:
        if r["p"] >= 1e-7:
            assert r["fast"] <= 1e-6 * r["p"] and r["exact"] <= 1e-6 * r["p"], r
    # 2. below: every difference is ABSOLUTE-small, decade after decade — the signature of 1 - integral
    worst_dev = max(max(r["fast"], r["exact"]) for r in rows)
    worst_self = max(r["self"] for r in rows)
    assert worst_dev <= 5e-13, worst_dev
    # 3. ... and the oracle moves by the same order of magnitude against ITSELF on an equivalent input: two correct
    #    evaluations of the reference's formulas differ that much (if it did not move at all the gap would be the device's)
    small = [r for r in rows if r["p"] < 1e-8]
    assert small
    assert worst_self > 0.0
    assert max(r["self"] for r in small) >= 0.02 * max(max(r["fast"], r["exact"]) for r in small), (worst_self, worst_dev)
    # 4. exact Davies does not close the gap: it is not the product form either
    assert max(r["exact"] for r in small) >= 0.05 * max(r["fast"] for r in small) or max(r["fast"] for r in small) < 1e-15
