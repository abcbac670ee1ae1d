"""CPU: Davies / Liu — oracle restatement and device algorithm (host harness) against the COMPILED REFERENCE
fragment.  The committed fixture tests/golden/davies_liu.json holds the reference's outputs so the check also
runs where /root/reference is absent; when oracle/_ref is present the live library is compared too."""
import json
import os

import numpy as np
import pytest

import hc
import orc

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "davies_liu.json")))["cases"]


def test_known_answers_of_reference_test():
    """regression/test/testMixtureChiSquare.cpp:13-38 — values printed by the reference code itself."""
    got = [(c["davies"], c["liu"]) for c in GOLD[:3]]
    assert abs(got[0][0] - 0.55106) < 5e-6 and abs(got[0][1] - 0.539917) < 5e-7
    assert abs(got[1][0] - 1.3959e-06) < 5e-11 and abs(got[1][1] - 1.38006e-06) < 5e-12
    assert got[2][0] == 0.0 and abs(got[2][1] - 7.9892e-11) < 5e-16


def test_davies_bit_exact_and_liu_close():
    nd = 0
    for c in GOLD:
        lam, Q = np.array(c["lambda"]), c["Q"]
        p, fault, nt = hc.davies(lam, Q)
        if len(lam) == 1:  # getPvalue delegates to Liu for a single coefficient
            assert abs(orc.davies(lam, Q) - c["davies"]) <= 1e-11 * abs(c["davies"]) + 2e-15
            assert abs(p - c["davies"]) <= 1e-11 * abs(c["davies"]) + 2e-15
            continue
        # oracle restatement: Davies bit-for-bit the reference (same operation order, same libm)
        assert orc.davies(lam, Q) == c["davies"], c
        if Q >= 0:
            assert p == c["davies"], c
            p2, _, _ = hc.davies(lam, Q, cached=True)
            assert p2 == p
        else:
            # device shortcut for Q < 0 (all coefficients > 0): the reference returns 1 or the fault marker -1,
            # both of which every hot-path caller replaces by Liu's value
            assert p == 1.0 and c["davies"] in (1.0, -1.0)
        # Liu goes through a different incomplete-gamma implementation: equal to ~1e-13, absolute floor because the
        # non-central tail is 0.5 + (0.5 - cdf) in the reference itself
        for got in (orc.liu(lam, Q), hc.liu(lam, Q)):
            assert abs(got - c["liu"]) <= 1e-11 * abs(c["liu"]) + 2e-15, (got, c["liu"])
        nd += 1
    assert nd > 400


def test_negative_q_is_one_or_fault():
    """The invariant behind the device's Q < 0 shortcut, checked on the live reference when it is available."""
    if orc.ref() is None:
        pytest.skip("oracle/_ref not built (no /root/reference here); covered by the committed fixture")
    rng = np.random.default_rng(5)
    for _ in range(3000):
        r = int(rng.integers(2, 90))
        lam = np.sort(rng.gamma(rng.choice([0.3, 1, 5]), 1.0, size=r) * 10 ** rng.uniform(-8, 8))[::-1].copy()
        c = -lam.sum() * 10 ** rng.uniform(-16, 6)
        assert orc.davies(lam, c, "ref") in (1.0, -1.0)


def test_oracle_matches_live_reference():
    if orc.ref() is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(9)
    for _ in range(1500):
        r = int(rng.integers(2, 80))
        lam = np.sort(rng.gamma(0.5, 1.0, size=r) * rng.choice([1, 1e-3, 10]))[::-1].copy()
        Q = lam.sum() * rng.choice([0.01, 0.1, 0.5, 1, 2, 5, 10, 50]) * rng.uniform(0.5, 1.5)
        assert orc.davies(lam, Q) == orc.davies(lam, Q, "ref")
        a, b = orc.liu(lam, Q), orc.liu(lam, Q, "ref")
        assert abs(a - b) <= 1e-11 * abs(b) + 2e-15
        v1, f1, t1 = orc.qf(lam, Q)
        v2, f2, t2 = orc.qf(lam, Q, "ref")
        assert v1 == v2 and f1 == f2 and np.array_equal(t1, t2)


def test_product_form_davies_matches_term_by_term():
    """The engine's default evaluation of qf()'s coefficient sums (complex / real products, one atan2 / log per sum,
    rvt_davies.h) against the term-by-term evaluation that is bit-identical to the reference: same number of integrand
    terms (the searches took the same path), p-values equal to ~1e-15 absolute."""
    rng = np.random.default_rng(11)
    cases = [(np.array(c["lambda"]), c["Q"]) for c in GOLD if len(c["lambda"]) > 1 and c["Q"] >= 0 and
             min(c["lambda"]) > 0]
    for _ in range(600):
        r = int(rng.integers(2, 96))
        lam = np.sort(rng.gamma(rng.choice([0.3, 0.5, 1, 5]), 1.0, size=r) * 10 ** rng.uniform(-6, 6))[::-1].copy()
        lam = lam[lam > 0]
        if len(lam) < 2:
            continue
        Q = lam.sum() * rng.choice([0.01, 0.1, 0.5, 1, 2, 5, 10, 30, 80]) * rng.uniform(0.5, 1.5)
        cases.append((lam, Q))
    for _ in range(300):      # coefficients of either sign (not met on the hot path; the product form still applies)
        r = int(rng.integers(2, 60))
        lam = rng.gamma(0.7, 1.0, size=r) * rng.choice([-1.0, 1.0], size=r, p=[0.3, 0.7]) * 10 ** rng.uniform(-3, 3)
        Q = np.abs(lam).sum() * rng.choice([-1, -0.2, 0.05, 0.5, 1, 3, 10]) * rng.uniform(0.5, 1.5)
        if Q >= 0:
            cases.append((lam, Q))
    worst_abs = worst_rel = 0.0
    n_terms_differ = 0
    for lam, Q in cases:
        for cached in (False, True):
            pe, fe, nte = hc.davies(lam, Q, cached=cached)
            pf, ff, ntf = hc.davies(lam, Q, cached=cached, fast=True)
            assert fe == ff
            n_terms_differ += nte != ntf
            if fe == 0:
                worst_abs = max(worst_abs, abs(pe - pf))
                if pe > 1e-6:
                    worst_rel = max(worst_rel, abs(pe - pf) / pe)
    assert n_terms_differ == 0
    assert worst_abs <= 5e-15 and worst_rel <= 1e-8, (worst_abs, worst_rel)   # observed: 1.6e-15 / 1e-9


# ---- the reference's GenotypeCounter / SNPHWE / RingMemoryPool compiled where they lie (oracle/_ref/libref_counter.so) ----
def _counter_columns(rng, N, M):
    """Columns that hit every branch of GenotypeCounter::add: hard calls, missing (< 0), dosages around the 2/3 and 4/3
    thresholds, exactly 2.0, above 2.0 (counted missing but still in nSample)."""
    G = rng.choice([0.0, 1.0, 2.0], size=(N, M), p=[0.8, 0.15, 0.05])
    G[rng.random((N, M)) < 0.03] = -9.0
    dos = rng.random((N, M)) < 0.1
    G[dos] = np.round(rng.uniform(0, 2.2, size=dos.sum()), 3)
    G[0, :4] = [2.0 / 3, 4.0 / 3, 2.0, 2.0000001]
    G[1, :4] = [np.nextafter(2.0 / 3, 0), np.nextafter(4.0 / 3, 0), np.nextafter(2.0, 3), -0.0]
    G[:, M - 1] = -9.0                       # an all-missing column: AF = 0 with nSample = N
    return np.asfortranarray(G)


def test_counter_af_against_the_compiled_reference_counter():
    """orc_counter_af (the weights of every kernel test, quirk #4) vs src/GenotypeCounter.h as compiled."""
    R = orc.ref_counter()
    if R is None:
        pytest.skip("oracle/_ref/libref_counter.so not built (no /root/reference here)")
    rng = np.random.default_rng(11)
    for N, M in ((1, 3), (7, 5), (500, 12), (4001, 9)):
        G = _counter_columns(rng, max(N, 2), max(M, 5))[:N, :M].copy(order="F") if N >= 2 else np.asfortranarray(
            rng.choice([0.0, 1.0, 2.0, -9.0], size=(N, M)))
        af = orc.counter_af(G)
        for j in range(M):
            out = np.zeros(8)
            col = np.ascontiguousarray(G[:, j])
            R.ref_counter(orc._dp(col), N, orc._dp(out))
            assert af[j] == out[0], (N, j, af[j], out[0])       # same additions in the same order: bit-identical
            ok = (col >= 0) & (col <= 2.0)
            assert out[7] == N - ok.sum() and out[4] + out[5] + out[6] == ok.sum()
            assert 0.0 <= out[3] <= 1.0


def test_hwe_exact_test_known_values():
    """SNPHWE through GenotypeCounter::getHWE: the all-missing convention and a textbook table."""
    R = orc.ref_counter()
    if R is None:
        pytest.skip("oracle/_ref/libref_counter.so not built")
    out = np.zeros(8)
    col = np.full(10, -9.0)
    R.ref_counter(orc._dp(col), 10, orc._dp(out))
    assert out[3] == 0.0 and out[0] == 0.0 and out[2] == 0.0
    # 57 het / 14 hom-alt / 50 hom-ref ... in HWE proportions the p-value is large; a het deficit gives a small one
    col = np.array([1.0] * 42 + [0.0] * 49 + [2.0] * 9)
    R.ref_counter(orc._dp(col), len(col), orc._dp(out))
    assert out[3] > 0.9
    col = np.array([1.0] * 2 + [0.0] * 69 + [2.0] * 29)
    R.ref_counter(orc._dp(col), len(col), orc._dp(out))
    assert out[3] < 1e-15


def test_metacov_window_walk_on_the_reference_ring_pool():
    """MetaCovTest keeps its window in RingMemoryPool chunks that are allocated per kept variant and released in queue
    order (src/Model.h:3956-3990, quirk #20).  Drive the same walk over the REAL pool — chunk payload = the variant's
    index, so a stale or moved chunk shows — and require the (head, marker) pairs it visits to be exactly the finite
    entries of orc_metacov's band, through several doublings of the pool."""
    R = orc.ref_counter()
    if R is None:
        pytest.skip("oracle/_ref/libref_counter.so not built")
    rng = np.random.default_rng(3)
    N, V, d = 60, 300, 2
    G = rng.choice([0.0, 1.0, 2.0], size=(N, V), p=[0.7, 0.25, 0.05])
    mono = rng.random(V) < 0.1
    G[:, mono] = 0.0
    chrom = np.repeat([1, 2], V // 2).astype(np.int32)
    pos = np.sort(rng.integers(0, 4000, size=V)).astype(np.int32)
    pos[V // 2:] -= pos[V // 2]
    X = np.column_stack([np.ones(N), rng.normal(size=N)])
    y = rng.normal(size=N)
    window = 700                                   # > 64 kept variants in a window: the pool has to grow (64 -> 128 -> ...)
    rc, kept, cov, row_end, xz, zz = orc.metacov(G, chrom, pos, X, y, 0, window)
    assert rc == 0
    pool = R.ref_ring_new(4, 2)                    # small start: growth while head > tail is exercised (case 2 / 4)
    queue, pairs, grew = [], set(), 0
    try:
        def evict_head():
            h, idx = queue.pop(0)
            for (j, jdx) in [(h, idx)] + queue:
                assert R.ref_ring_chunk(pool, jdx)[0] == float(j)     # the chunk of index jdx still holds variant j
                pairs.add((h, j))
            R.ref_ring_deallocate(pool, idx)
        for j in range(V):
            while queue and (chrom[queue[0][0]] != chrom[j] or abs(int(pos[j]) - int(pos[queue[0][0]])) > window):
                evict_head()
            if not kept[j]:
                continue
            cap = R.ref_ring_capacity(pool)
            idx = R.ref_ring_allocate(pool)
            grew += R.ref_ring_capacity(pool) > cap
            R.ref_ring_chunk(pool, idx)[0] = float(j)
            queue.append((j, idx))
        while queue:
            evict_head()
        assert R.ref_ring_size(pool) == 0
    finally:
        R.ref_ring_delete(pool)
    assert grew >= 4
    want = {(h, j) for h in range(V) for j in range(h, V) if np.isfinite(cov[h, j])}
    assert pairs == want and len(want) > 5000
    assert all((row_end[h] == max(j for (hh, j) in want if hh == h)) for h in range(V) if kept[h])


def test_search_memo_changes_nothing_and_stays_small():
    """rvt_davies.h DaviesMemo: the coefficient sums of errbd / truncation depend on the evaluation point alone and the
    searches of qf() visit a small lattice of points whatever the quantile; a hit must return the doubles a miss computes
    (bit-identical p-values with and without the memo) and hundreds of quantiles — over a far wider range than one
    SKAT-O quadrature asks for — must typically need a dozen slots of the 32- / 64-slot tables (a full table only stops
    memoising: coefficient sets of two or three terms that repeat the auxiliary integration can fill it)."""
    rng = np.random.default_rng(21)
    used = []
    for trial in range(40):
        r = int(rng.integers(2, 96))
        lam = np.sort(rng.gamma(rng.choice([0.3, 1, 5]), 1.0, size=r) * 10 ** rng.uniform(-6, 6))[::-1].copy()
        mu, sd = lam.sum(), np.sqrt(2 * (lam ** 2).sum())
        Qs = np.concatenate([mu + sd * rng.uniform(-1.5, 12, size=300), mu * 10 ** rng.uniform(-3, 1.5, size=100), [0.0]])
        p_memo, slots = hc.davies_memo_sweep(lam, Qs)
        for Q, pm in zip(Qs[::7], p_memo[::7]):
            p0, _, _ = hc.davies(lam, Q, cached=True, fast=True)
            assert p0 == pm or (np.isnan(p0) and np.isnan(pm)), (trial, Q, p0, pm)
        used.append(slots)
    used = np.array(used)
    assert np.median(used[:, 0]) <= 16 and np.median(used[:, 1]) <= 16, used.T   # (observed: 10-15 and 6-16)
    assert used[:, 0].max() <= 32 and used[:, 1].max() <= 64
