"""CPU: the oracle's scalar functions AND the device algorithms (host harness build of the RVT_HD code) against
the GSL 1.16 golden vectors (tests/golden/gsl_scalar.json, generated from the tarball the reference vendors)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import hc
import orc

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gsl_scalar.json")))["cases"]


def _rel(a, b):
    return abs(a - b) / abs(b) if b != 0 else abs(a)


@pytest.mark.parametrize("op", ["beta", "chisqQ", "chisqP", "chisqQinv", "chisqpdf"])
def test_special_functions_match_gsl(op):
    L, H = orc.lib(), hc.lib()
    fo = {"beta": L.orc_beta_pdf, "chisqQ": L.orc_chisq_Q, "chisqP": L.orc_chisq_P, "chisqQinv": L.orc_chisq_Qinv,
          "chisqpdf": L.orc_chisq_pdf}[op]
    fh = {"beta": H.hc_beta_pdf, "chisqQ": H.hc_chisq_Q, "chisqP": H.hc_chisq_P, "chisqQinv": H.hc_chisq_Qinv,
          "chisqpdf": H.hc_chisq_pdf}[op]
    n = 0
    for c in GOLD:
        if c["op"] != op:
            continue
        ref = c["value"]
        assert _rel(fo(*c["args"]), ref) < 2e-12, (op, c["args"])
        assert _rel(fh(*c["args"]), ref) < 2e-12, (op, c["args"])
        n += 1
    assert n > 50


def test_qags_matches_gsl():
    """Same abscissae => same number of intervals; results equal to rounding (bit-equal for libm-only integrands)."""
    L, H = orc.lib(), hc.lib()
    n = 0
    for c in GOLD:
        if c["op"] != "qags":
            continue
        a = c["args"]
        for fn in (L.orc_qags_builtin, H.hc_qags_builtin):
            r, e, ne = C.c_double(), C.c_double(), C.c_int()
            st = fn(int(a[0]), a[1], a[2], a[3], a[4], a[5], int(a[6]), C.byref(r), C.byref(e), C.byref(ne))
            assert st == c["status"]
            assert ne.value == 21 * (2 * c["intervals"] - 1)
            assert abs(r.value - c["result"]) <= 1e-12 * abs(c["result"])
            # the error ESTIMATE amplifies 1e-15 differences of the integrand (lgamma) when it is itself ~1e-13
            assert abs(e.value - c["abserr"]) <= 1e-6 * abs(c["abserr"]) + 1e-13
        n += 1
    assert n >= 10


def test_oracle_vs_scipy():
    from scipy import stats
    rng = np.random.default_rng(0)
    L = orc.lib()
    for _ in range(500):
        df = float(rng.uniform(0.5, 60))
        x = float(df * 10 ** rng.uniform(-2, 1))
        assert _rel(L.orc_chisq_Q(x, df), stats.chi2.sf(x, df)) < 1e-11
        q = float(rng.uniform(1e-10, 0.99))
        assert _rel(L.orc_chisq_Qinv(q, df), stats.chi2.isf(q, df)) < 1e-9


def test_glibc_rand_emulator_matches_libc():
    libc = C.CDLL("libc.so.6")
    L = orc.lib()
    for seed in (1, 12345):
        libc.srand(seed)
        L.orc_rand_seed(seed)
        assert all(libc.rand() == L.orc_rand() for _ in range(20000))


def test_eigen_solvers_agree_with_lapack():
    rng = np.random.default_rng(3)
    for n in (1, 2, 5, 17, 50, 96):
        B = rng.normal(size=(n + 3, n))
        A = B.T @ B
        if n > 4:
            A[:, 1] = A[:, 0]
            A[1, :] = A[0, :]  # rank deficient
        w0 = np.linalg.eigvalsh(A)
        scale = max(abs(w0).max(), 1e-300)
        assert np.max(np.abs(orc.sym_eigvals(A) - w0)) < 1e-12 * scale
        assert np.max(np.abs(hc.sym_eigvals(A) - w0)) < 1e-12 * scale


def test_division_free_sturm_bisection_on_hard_tridiagonals():
    """rvt_coop.h's product-form Sturm count (rescaled every four rows) against LAPACK on the matrices that break a naive
    product recurrence: exactly split matrices, graded ones (under- / overflow of the unscaled sequence within a few
    rows), huge and tiny scales, clustered and repeated eigenvalues, Wilkinson's W21+."""
    import scipy.linalg as sla
    rng = np.random.default_rng(11)
    cases = []
    n = 64
    cases.append((rng.normal(size=n), rng.normal(size=n - 1)))
    d, e = rng.normal(size=n), rng.normal(size=n - 1)
    e[[7, 8, 30]] = 0.0                                            # exact splits, two of them adjacent
    cases.append((d, e))
    cases.append((np.zeros(n), np.zeros(n - 1)))                  # the zero matrix
    cases.append((np.ones(n), np.zeros(n - 1)))                   # n-fold eigenvalue, fully split
    cases.append((10.0 ** np.linspace(150, -150, n), 10.0 ** np.linspace(149, -149, n - 1)))   # graded over 300 decades
    cases.append((rng.normal(size=n) * 1e150, rng.normal(size=n - 1) * 1e150))
    cases.append((rng.normal(size=n) * 1e-150, rng.normal(size=n - 1) * 1e-150))
    cases.append((np.abs(np.arange(-10, 11)).astype(float), np.ones(20)))                      # W21+
    cases.append((np.full(n, 2.0), np.full(n - 1, -1.0)))         # second difference: the sequence grows like 3^j unscaled
    cases.append((np.full(200, 1e5), np.full(199, 1e-3)))         # long, nearly diagonal (p shrinks by 1e-8 per row at x = d)
    d, e = rng.normal(size=n), rng.normal(size=n - 1) * 1e-12     # clustered pairs
    d[1::2] = d[0::2]
    cases.append((d, e))
    cases.append((np.array([3.0]), np.zeros(0)))
    cases.append((np.array([1.0, 1.0]), np.array([1e-200])))
    # a positive semi-definite spectrum spanning 1e-15 .. 1 (what SKAT's eigenvalue cut at 1e-30 and the Davies inputs see),
    # once coupled, once with exact splits, once with diagonal entries that ARE eigenvalues (d_j - x = 0 exactly at a
    # bisection point: a zero member of the Sturm sequence, whose sign bit the count reads — ADVICE r4)
    lam = 10.0 ** np.linspace(-15, 0, 48)
    Q, _ = np.linalg.qr(rng.normal(size=(48, 48)))
    T = sla.hessenberg((Q * lam) @ Q.T)
    d, e = np.diag(T).copy(), np.diag(T, 1).copy()
    cases.append((d, e))
    e2 = e.copy()
    e2[[5, 6, 20, 40]] = 0.0
    cases.append((d, e2))
    cases.append((np.array([0.0, 0.5, 0.5, 0.25, 1.0, 0.0]), np.array([0.0, 0.0, 0.25, 0.0, 0.0])))
    cases.append((np.array([0.5, 0.5, 0.5]), np.array([0.25, 0.25])))      # x = 0.5 = d_0 at the first bisection step
    for d, e in cases:
        w0 = sla.eigvalsh_tridiagonal(d, e) if len(d) > 1 else np.array(d)
        w = hc.tridiag_eigvals(d, e)
        scale = max(np.abs(w0).max(), 1e-300)
        assert np.all(np.isfinite(w))
        assert np.all(np.diff(w) >= 0)
        assert np.max(np.abs(w - w0)) <= 4e-15 * scale * max(1, len(d) / 16) + 1e-290, (len(d), np.max(np.abs(w - w0)) / scale)
