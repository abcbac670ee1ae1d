"""SKAT permutation p-values through the C ABI.  Exact mode (rvt_set_perm_exact): the same emulated glibc rand() stream as
the oracle, so the permutations — and with them ActualPerm / NumGreater / NumEqual — are identical, gene after gene.
Counter-based mode (the default): other random numbers, the same estimator — compared with the exact mode within binomial
error."""
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


# (300 x 0.0333 x 2 = 19.98: the reference's threshold is an INT member, src/Permutation.h:153 -> 19, not 19.98 -> a stop one
#  hit earlier; 100 x 0.001 x 2 = 0.2 -> 0: no permutation at all, p = 1 — both pinned on the reference's compiled class in
#  tests/test_oracle_ref.py)
@pytest.mark.parametrize("N,n_perm,alpha,explicit", [(331, 300, 0.05, True), (1000, 120, 0.2, True), (331, 300, 0.05, False),
                                                     (331, 300, 0.0333, True), (331, 100, 0.001, True)])
def test_permutation_counts_match_oracle(eng, N, n_perm, alpha, explicit):
    """explicit = False: NO mode is selected — a fresh single context must reproduce the reference's counters by DEFAULT
    (the counter-based mode is for genes dealt over several devices, or on request)."""
    import rvtests_amd
    if not explicit:
        eng = rvtests_amd.Engine(0)     # a context nobody has configured
    d = 2
    genes = [synth.make_gene(N, M, seed=500 + M, missing=0.01, common=True, mono=True)[1:] for M in (8, 1, 21, 5)]
    genes.insert(2, (np.zeros((N, 3)), np.zeros(3)))       # no polymorphic column: no permutations, no draws
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=8, G_effect=0.8 * genes[0][0][:, :2].sum(1))
    eng.set_null(0, X, res, v, s2)
    prm = rvtests_amd.Params(1.0, 25.0, 1.0, 25.0, n_perm, alpha)
    ptrs = [eng.upload_block(G) for G, af in genes]
    if explicit:
        eng.set_perm_exact(True)
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
    if n_perm == 100:                  # threshold truncated to 0: Permutation::next() is false before the first shuffle
        assert all(r.perm_actual_perm == 0 and r.perm_pvalue == 1.0 for r in out if r.perm_ok)


def test_counter_based_mode_agrees_with_the_exact_mode_within_binomial_error(eng):
    """200 null genes, N = 2000: permutation p-values of the counter-based mode against the replay of the reference's
    stream — two Monte-Carlo estimates of the same tail probability, so their difference scales with the binomial
    standard errors; and against the analytic SKAT p-value (both modes estimate it)."""
    import rvtests_amd
    N, d, n_genes, n_perm, alpha = 2000, 2, 200, 2000, 0.05
    rng = np.random.default_rng(99)
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=21)
    eng.set_null(0, X, res, v, s2)
    genes = []
    for g in range(n_genes):
        M = int(rng.integers(3, 40))
        G = np.asfortranarray(rng.binomial(2, 10 ** rng.uniform(-2.3, -0.8, M), size=(N, M)).astype(np.float64))
        genes.append((G, G.sum(0) / (2.0 * N)))
    ptrs = [eng.upload_block(G) for G, af in genes]
    prm = rvtests_amd.Params(1.0, 25.0, 1.0, 25.0, n_perm, alpha)
    Ms, afs = [G.shape[1] for G, af in genes], [af for G, af in genes]
    eng.set_perm_exact(True)
    eng.rand_seed(1)
    exact = eng.run_blocks(ptrs, Ms, afs, tests=rvtests_amd.TEST_SKAT, params=prm)
    eng.set_perm_exact(False)
    eng.rand_seed(1)
    cb = eng.run_blocks(ptrs, Ms, afs, tests=rvtests_amd.TEST_SKAT, params=prm, ids=list(range(100, 100 + n_genes)))
    cb2 = eng.run_blocks(ptrs[::-1], Ms[::-1], afs[::-1], tests=rvtests_amd.TEST_SKAT, params=prm,
                         ids=list(range(100, 100 + n_genes))[::-1])[::-1]
    z = []
    for a, b, b2 in zip(exact, cb, cb2):
        assert a.perm_ok and b.perm_ok
        # keyed by gene id: the order of the genes does not matter
        assert (b.perm_actual_perm, b.perm_num_greater, b.perm_pvalue) == (b2.perm_actual_perm, b2.perm_num_greater, b2.perm_pvalue)
        pa, pb = a.perm_pvalue, b.perm_pvalue
        se = np.sqrt(pa * (1 - pa) / a.perm_actual_perm + pb * (1 - pb) / b.perm_actual_perm) + 1e-9
        z.append((pb - pa) / se)
    z = np.array(z)
    assert abs(z.mean()) < 0.25 and 0.7 < z.std() < 1.35 and (np.abs(z) > 4).sum() == 0, (z.mean(), z.std(), np.abs(z).max())
    # both estimate the analytic p-value (Davies): no systematic offset of the counter-based estimates
    dev = np.array([(b.perm_pvalue - b.skat_p) for b in cb])
    assert abs(dev.mean()) < 0.02
    for p in ptrs:
        eng.free_block(p)


@pytest.mark.parametrize("exact", [True, False])
def test_a_context_whose_null_model_grows_keeps_no_buffer_of_the_old_size(eng, exact):
    """One context, two analyses: N = 700 with 400 permutations, then N = 9 001 with 30 (fewer shuffles of more samples: the
    chunk of shuffles N x B shrinks while every per-sample buffer grows).  The second analysis gives what a fresh context gives
    (round 6: the flipped-genotype buffer of the permutation stage was kept by its column count alone and overran)."""
    import rvtests_amd
    outs = []
    for fresh in (False, True):
        e = rvtests_amd.Engine(0) if fresh else eng
        for N, n_perm in ((700, 400), (9001, 30)) if not fresh else ((9001, 30),):
            genes = [synth.make_gene(N, M, seed=900 + M, missing=0.01, common=True, mono=True)[1:] for M in (24, 7, 40)]
            X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=18, G_effect=0.5 * genes[0][0][:, :2].sum(1))
            e.set_null(0, X, res, v, s2)
            prm = rvtests_amd.Params(1.0, 25.0, 1.0, 25.0, n_perm, 0.4)
            e.set_perm_exact(exact)
            if exact:
                e.rand_seed(1)
            ptrs = [e.upload_block(G) for G, af in genes]
            out = e.run_blocks(ptrs, [G.shape[1] for G, af in genes], [af for G, af in genes], tests=rvtests_amd.TEST_SKAT,
                               params=prm, ids=[0, 1, 2])   # (the counter-based shuffles are a function of the gene id)
            for p in ptrs:
                e.free_block(p)
        outs.append([(r.skat_Q, r.skat_p, r.perm_actual_perm, r.perm_num_greater, r.perm_num_equal, r.perm_pvalue) for r in out])
        if fresh:
            e.close()
    assert outs[0] == outs[1] and all(t[2] > 0 for t in outs[0])
