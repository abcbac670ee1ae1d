"""CPU: the oracle's KBAC restatement against vectors produced by the REFERENCE's own kbac.cpp + GSL 1.16
(tests/golden/kbac.json, generator committed): p-values bit-equal, the position of the process-wide rand() stream after
the test equal, gsl_cdf_hypergeometric_P bit-equal."""
import json
import os

import numpy as np

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "kbac.json")))


def test_hypergeometric_cdf_matches_gsl():
    worst = 0.0
    for k, n1, n2, t, want in GOLD["hypergeometric_P"]:
        got = orc.hypergeometric_P(k, n1, n2, t)
        worst = max(worst, abs(got - want) / max(abs(want), 1e-300))
        assert got == want or abs(got - want) <= 4e-16 * abs(want), (k, n1, n2, t, got, want)
    assert worst <= 4e-16


def test_kbac_matches_reference_vectors():
    for c in GOLD["cases"]:
        G = np.array(c["G"])
        p, obs, npat, done = orc.kbac(G, np.array(c["y"]), np.array(c["maf"]), c["nperm"], c["alpha"], seed=c["seed"])
        nxt = orc.lib().orc_rand()
        assert p == c["pvalue"], (c["N"], c["M"], p, c["pvalue"])
        assert nxt == c["next_rand"]
