"""SKAT permutation p-values through the C ABI against the oracle: same emulated glibc rand() stream, so the
permutations — and with them ActualPerm / NumGreater / NumEqual — are identical, gene after gene."""
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


def test_glibc_stream_matches_libc():
    """the oracle's generator is glibc's: compare with the C library itself"""
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    orc.rand_seed(1)
    assert [libc.rand() for _ in range(2000)] == [orc.lib().orc_rand() for _ in range(2000)]


@pytest.mark.parametrize("N,n_perm,alpha", [(331, 300, 0.05), (1000, 120, 0.2)])
def test_permutation_counts_match_oracle(eng, N, n_perm, alpha):
    import rvtests_amd
    d = 2
    genes = [synth.make_gene(N, M, seed=500 + M, missing=0.01, common=True, mono=True)[1:] for M in (8, 1, 21, 5)]
    genes.insert(2, (np.zeros((N, 3)), np.zeros(3)))       # no polymorphic column: no permutations, no draws
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=8, G_effect=0.8 * genes[0][0][:, :2].sum(1))
    eng.set_null(0, X, res, v, s2)
    prm = rvtests_amd.Params(1.0, 25.0, 1.0, 25.0, n_perm, alpha)
    ptrs = [eng.upload_block(G) for G, af in genes]
    eng.rand_seed(1)
    out = eng.run_blocks(ptrs, [G.shape[1] for G, af in genes], [af for G, af in genes],
                         tests=rvtests_amd.TEST_SKAT, params=prm)
    orc.rand_seed(1)
    stopped_early = 0
    for r, (G, af) in zip(out, genes):
        rc, a = orc.skat(G, af, X, res, v, 0)
        if a.n_poly == 0:
            assert r.skat_ok == 0 and r.perm_ok == 0
            continue
        assert abs(r.skat_Q - a.Q) <= 1e-10 * a.Q
        # the oracle permutes in fp64 with the device's observed statistic as threshold reference
        rc, p = orc.skat_permute(G, af, res, r.skat_Q, n_perm, alpha)
        assert rc == 0
        assert r.perm_ok == 1 and r.perm_num_perm == n_perm
        assert (r.perm_actual_perm, r.perm_num_greater, r.perm_num_equal) == (p.actual_perm, p.num_x, p.num_equal)
        assert r.perm_pvalue == p.pvalue
        stopped_early += p.actual_perm < n_perm
    assert stopped_early >= 1          # the adaptive stop was exercised
