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


# ---- the reference's Permutation stop rule, ModelParser and floatToString compiled where they lie (oracle/_ref/libref_host.so) ----
def _ref_host():
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_host.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_host.so not built (needs the reference tree: make -C oracle ref)")
    L = C.CDLL(path)
    L.ref_permutation_run.restype = C.c_double
    L.ref_permutation_run.argtypes = [C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    L.ref_parser_parse.restype = C.c_int
    L.ref_parser_parse.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.ref_parser_has.argtypes = [C.c_char_p]
    L.ref_parser_value.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.ref_parser_double.restype = C.c_double
    L.ref_parser_double.argtypes = [C.c_char_p, C.c_double]
    L.ref_parser_int.argtypes = [C.c_char_p, C.c_int]
    L.ref_parser_bool.argtypes = [C.c_char_p, C.c_int]
    L.ref_float_to_string.argtypes = [C.c_double, C.c_char_p, C.c_int]
    L.ref_float_to_string_f32.argtypes = [C.c_float, C.c_char_p, C.c_int]
    return L


def test_permutation_stop_rule_matches_the_compiled_reference():
    """The oracle's adaptive stop rule (orc_perm_stop_run = the PermStop that orc_skat_permute runs, which the engine's counters
    are compared with in tests/test_gpu_perm.py) against the reference's own Permutation class (src/Permutation.h:48-158) on
    the same sequences of statistics: permutations used, counters and p-value identical — early stops, ties with the observed
    value, nPerm exhausted, alpha = 0 and an empty list included."""
    import ctypes as C
    R = _ref_host()
    O = orc.lib()
    O.orc_perm_stop_run.restype = C.c_double
    O.orc_perm_stop_run.argtypes = R.ref_permutation_run.argtypes
    rng = np.random.default_rng(20)
    n_early = 0
    for trial in range(400):
        nperm = int(rng.choice([0, 1, 10, 100, 1000, 10000]))
        alpha = float(rng.choice([0.0, 0.001, 0.05, 0.5, 1.0]))
        obs = float(rng.normal())
        n = int(rng.integers(0, nperm + 5))
        stats = rng.normal(size=max(n, 1)) + rng.choice([-1.0, 0.0, 1.5])
        stats[rng.random(max(n, 1)) < 0.05] = obs                 # ties
        sp = stats.ctypes.data_as(C.POINTER(C.c_double))
        a, b = (C.c_int * 3)(), (C.c_int * 3)()
        pr = R.ref_permutation_run(nperm, alpha, obs, sp, n, a)
        po = O.orc_perm_stop_run(nperm, alpha, obs, sp, n, b)
        assert list(a) == list(b) and pr == po, (trial, nperm, alpha, list(a), list(b))
        n_early += a[0] < min(n, nperm)
    assert n_early > 50                                            # the rule did stop early in many trials


def _driver(args, stdin=None):
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rvtests_amd", "csrc", "host", "host_driver")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    p = subprocess.run([exe] + args, input=stdin, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    return p.stdout.splitlines()


def test_model_parser_matches_the_compiled_reference():
    """The adapters' ModelParser (csrc/host/ModelFitterGpu.cpp) against the reference's compiled src/ModelParser.cpp: return code,
    model name, number of parameters, hasTag / value and the assign() conversions (int through a double: "1e4" is 10 000;
    case folding; ':' and ',' separators; flags without a value; empty pieces; a missing ']')."""
    import ctypes as C
    R = _ref_host()
    specs = ["skat", "Skat[nPerm=1e4:alpha=0.05,beta1=1:beta2=25]", "SKATO", "cmc[]", "kbac[nPerm=10000,alpha=0.001]",
             "famSkat[Beta1=0.5]", "cov[windowSize=1000000:gwama]", "score[]", "skat[nperm=5", "vt[a=1::b=2]", "x[=3]", "x[a==3,b=]",
             "analytic[k=1.9,K=2.7]", "Skat[nPerm=-3.7]", "skat[nperm=abc]", "m[,]", "m[a]", "M[A=Hello:b=WORLD]", "[a=1]", "x]"]
    tags = ["nperm", "nPerm", "alpha", "beta1", "BETA2", "windowsize", "gwama", "a", "b", "k", "", "nosuch"]
    for spec in specs:
        name = C.create_string_buffer(256)
        npar = C.c_int()
        rc = R.ref_parser_parse(spec.encode(), name, 256, C.byref(npar))
        got = _driver(["--parse", spec] + tags)
        assert int(got[0]) == rc, spec
        if rc != 0:
            continue                                              # (a rejected spec: the models are not created)
        assert got[1] == name.value.decode() and int(got[2]) == npar.value, (spec, got[:3], name.value, npar.value)
        for tag, line in zip(tags, got[3:]):
            has, val, dv, iv, bv = line.split("\t")
            assert int(has) == R.ref_parser_has(tag.encode()), (spec, tag)
            buf = C.create_string_buffer(256)
            if R.ref_parser_value(tag.encode(), buf, 256):
                assert val == buf.value.decode(), (spec, tag)
            else:
                assert val == "<null>"
            assert float(dv) == R.ref_parser_double(tag.encode(), -7.5), (spec, tag)
            assert int(iv) == R.ref_parser_int(tag.encode(), -7), (spec, tag)
            assert int(bv) == R.ref_parser_bool(tag.encode(), 0), (spec, tag)


def test_float_to_string_matches_the_compiled_reference():
    """floatToString (base/TypeConversion.h:97-105: a stringstream at precision 6, what Result prints the CMC / Zeggini / covZZ
    numbers with) compiled from the reference against the adapters' floatToString and against C's "%g" (formatG, what the SKAT /
    MetaCov rows use): the three agree on every finite value tried, as doubles and through float."""
    import ctypes as C
    R = _ref_host()
    rng = np.random.default_rng(3)
    xs = [0.0, -0.0, 1.0, -1.0, 0.5, 1e-5, 9.99999e-5, 1e-4, 0.000123456789, 123456.0, 1234567.0, 999999.5, 1e6, 1e15, 1e16, 1e-300,
          5e-324, 1.7976931348623157e308, 0.1, 2.5e-7, 100.0, 1e5, 99999.95, 0.05, 3.14159265358979, 2.0 / 3.0]
    xs += list(10.0 ** rng.uniform(-320, 308, 3000) * rng.choice([-1.0, 1.0], 3000))
    xs += list(rng.normal(size=2000)) + list(np.round(rng.normal(size=500) * 1000, 2))
    for f32 in (False, True):
        if f32:
            vals = [float(np.float32(x)) for x in xs if abs(x) < 3e38]
            lines = _driver(["--format"], "".join("f:%r\n" % v for v in vals))
        else:
            vals = [float(x) for x in xs]
            lines = _driver(["--format"], "".join("%r\n" % v for v in vals))
        assert len(lines) == len(vals)
        buf = C.create_string_buffer(128)
        for v, line in zip(vals, lines):
            fts, g = line.split("\t")
            if f32:
                R.ref_float_to_string_f32(C.c_float(v), buf, 128)
            else:
                R.ref_float_to_string(v, buf, 128)
            ref = buf.value.decode()
            assert fts == ref, (v, fts, ref)
            assert g == "%g" % v == ref, (v, g, ref)               # (the stream at precision 6 prints what %g prints)
