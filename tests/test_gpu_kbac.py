"""GPU: KBAC through the C ABI (rvt_kbac_blocks) against the oracle, which is itself pinned bit-for-bit on vectors from
the reference's own kbac.cpp + GSL 1.16 (tests/test_kbac_cpu.py).  Everything is exact: the statistic, the permutation
counts, the p-value and — checked through a second gene that continues the stream — the position of the emulated
process-wide rand() stream."""
import json
import os

import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def eng():
    import rvtests_amd
    e = rvtests_amd.Engine(0)
    yield e
    e.close()


def _oracle(G, af, y, nperm, alpha):
    Gf, fl, kp = orc.flip_poly(G)
    m = Gf.shape[1]
    if m == 0:
        return None
    return orc.kbac(Gf, y, np.asarray(af)[:m], nperm, alpha)      # (filtered position j reads frequency j)


@pytest.mark.parametrize("N,Ms,nperm,alpha", [(700, (6, 15), 300, 0.05), (2100, (9, 30, 4), 5200, 0.01),
                                              (900, (12, 8), 400, 1.0)])
def test_kbac_matches_oracle_exactly(eng, N, Ms, nperm, alpha):
    rng = np.random.default_rng(N)
    y = (rng.random(N) < 0.4).astype(np.float64)
    X = np.ones((N, 1))
    eng.fit_null(1, X, y)                              # binary trait, intercept only (KBAC takes no covariates)
    genes = [synth.make_gene(N, M, seed=70 + M, missing=0.02, common=False, mono=(M > 8)) for M in Ms]
    if len(Ms) > 2:                                    # a gene with signal: the adaptive rule must not stop it early
        G = genes[1][1]
        carriers = (np.rint(G).sum(1) > 0)
        y = np.where(carriers & (rng.random(N) < 0.85), 1.0, y)
        eng.fit_null(1, X, y)
    ptrs = [eng.upload_block(g[1]) for g in genes]
    eng.rand_seed(11)
    out = eng.kbac_blocks(ptrs, [g[1].shape[1] for g in genes], [g[2] for g in genes], y, nperm, alpha)
    orc.rand_seed(11)
    for r, (af0, G, af) in zip(out, genes):
        o = _oracle(G, af, y, nperm, alpha)
        if o is None:
            assert r.fit_ok == 0
            continue
        p, obs, npat, done = o
        assert r.fit_ok == 1 and r.n_pattern == npat
        assert r.stat == obs
        assert r.pvalue == p
        assert r.actual_perm == min(done, nperm)


def test_kbac_reference_vectors_through_the_device(eng):
    """The reference's own vectors (tests/golden/kbac.json) through the device: same p-values."""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "kbac.json")))
    for c in gold["cases"]:
        G = np.array(c["G"])                           # already flipped / polymorphic as far as KBAC is concerned
        N, M = G.shape
        if (G.sum(0) > N).any() or any(len(np.unique(G[:, j])) < 2 for j in range(M)):
            continue                                   # (a column the engine itself would flip or drop)
        y = np.array(c["y"])
        eng.fit_null(1, np.ones((N, 1)), y)
        eng.rand_seed(c["seed"])
        r = eng.kbac_blocks([eng.upload_block(G)], [M], [np.array(c["maf"])], y, c["nperm"], c["alpha"])[0]
        assert r.pvalue == c["pvalue"], (N, M)
