"""CPU: the keyed bijection behind the counter-based SKAT permutations (rvtests_amd/csrc/perm_counter.h), through the host
test harness."""
import numpy as np


def test_counter_based_permutation_is_a_bijection_and_looks_uniform():
    """perm_counter.h through the host harness: pi_{g,s} is a bijection of [0, N) for awkward N, differs between shuffles
    and genes, and position i lands uniformly (chi-square over many shuffles)."""
    import ctypes as C
    import hc
    L = hc.lib()
    L.hc_perm_indices.restype = None
    L.hc_perm_indices.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]

    def perm(seed, gene, shuffle, n):
        out = np.empty(n, dtype=np.uint32)
        L.hc_perm_indices(seed, gene, shuffle, n, out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out

    for n in (2, 3, 5, 64, 65, 1000, 4099, 65536, 500000):
        p = perm(1, 7, 0, n)
        assert np.array_equal(np.sort(p), np.arange(n, dtype=np.uint32)), n
    a, b, c = perm(1, 7, 0, 4099), perm(1, 7, 1, 4099), perm(1, 8, 0, 4099)
    assert (a != b).mean() > 0.99 and (a != c).mean() > 0.99 and (perm(2, 7, 0, 4099) != a).mean() > 0.99
    # where does element 0 .. 9 go over 20 000 shuffles of n = 50?  each of the 50 positions 400 times on average
    n, S = 50, 20000
    counts = np.zeros((10, n))
    for s_ in range(S):
        p = perm(1, 3, s_, n)
        counts[np.arange(10), p[:10]] += 1
    chi2 = ((counts - S / n) ** 2 / (S / n)).sum(1)           # 49 degrees of freedom each
    assert chi2.max() < 110 and 30 < chi2.mean() < 70, chi2
    # pairs: P(pi(0) < pi(1)) = 1/2, and adjacent inputs are not mapped to adjacent outputs more often than chance
    lt = adj = 0
    for s_ in range(4000):
        p = perm(5, 11, s_, 1000).astype(np.int64)
        lt += p[0] < p[1]
        adj += (np.abs(np.diff(p)) == 1).sum()
    assert abs(lt - 2000) < 4 * np.sqrt(1000) and abs(adj / 4000 - 2.0) < 0.3      # E[#adjacent] = 2 (n - 1) / n


