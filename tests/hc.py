"""ctypes loader for the HOST TEST HARNESS of the device algorithms
(rvtests_amd/csrc/hostcheck.cpp -> librvt_hostcheck.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rvtests_amd", "csrc")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class Params(C.Structure):
    _fields_ = [("skat_beta1", C.c_double), ("skat_beta2", C.c_double), ("skato_beta1", C.c_double),
                ("skato_beta2", C.c_double), ("skat_nperm", C.c_int), ("skat_alpha", C.c_double)]


class GeneResult(C.Structure):
    _fields_ = [
        ("gene_id", C.c_int64), ("status", C.c_uint32), ("n_variants", C.c_int), ("n_poly", C.c_int),
        ("skat_ok", C.c_int), ("skat_Q", C.c_double), ("skat_p", C.c_double), ("skat_nlambda", C.c_int),
        ("skato_ok", C.c_int), ("skato_Q", C.c_double), ("skato_rho", C.c_double), ("skato_p", C.c_double),
        ("skato_qags_status", C.c_int), ("skato_qags_neval", C.c_int),
        ("cmc_ok", C.c_int), ("cmc_nonref", C.c_int), ("cmc_U", C.c_double), ("cmc_V", C.c_double),
        ("cmc_stat", C.c_double), ("cmc_p", C.c_double),
        ("zeg_ok", C.c_int), ("zeg_U", C.c_double), ("zeg_V", C.c_double), ("zeg_stat", C.c_double),
        ("zeg_p", C.c_double), ("davies_terms", C.c_double),
        ("perm_ok", C.c_int), ("perm_num_perm", C.c_int), ("perm_actual_perm", C.c_int),
        ("perm_num_greater", C.c_int), ("perm_num_equal", C.c_int), ("perm_pvalue", C.c_double),
        ("famskat_ok", C.c_int), ("famskat_Q", C.c_double), ("famskat_p", C.c_double),
        ("famcmc_ok", C.c_int), ("famcmc_af", C.c_double), ("famcmc_U", C.c_double), ("famcmc_V", C.c_double),
        ("famcmc_p", C.c_double),
        ("famzeg_ok", C.c_int), ("famzeg_af", C.c_double), ("famzeg_U", C.c_double), ("famzeg_V", C.c_double),
        ("famzeg_p", C.c_double),
        ("vt_ok", C.c_int), ("vt_optnum", C.c_int), ("vt_ncutoff", C.c_int),
        ("vt_minmaf", C.c_double), ("vt_maxmaf", C.c_double), ("vt_optmaf", C.c_double), ("vt_U", C.c_double),
        ("vt_V", C.c_double), ("vt_stat", C.c_double), ("vt_p", C.c_double), ("vt_p_error", C.c_double),
    ]


def default_params():
    return Params(1.0, 25.0, 1.0, 25.0, 0, 0.05)


def _dp(a):
    return a.ctypes.data_as(c_double_p)


_lib = None


def build():
    out = os.path.join(CSRC, "librvt_hostcheck.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", out,
                           os.path.join(CSRC, "hostcheck.cpp"), "-lm", "-lpthread"])
    return out


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("RVT_HOSTCHECK_LIB")     # (a caller-built variant, e.g. tools/davies_divergence.py's profiling build)
        if not path:
            path = os.path.join(CSRC, "librvt_hostcheck.so")
            srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".cpp"))]
            if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
                build()
        L = C.CDLL(path)
        d = C.c_double
        for n in ("hc_chisq_Q", "hc_chisq_P", "hc_chisq_Qinv", "hc_chisq_pdf"):
            getattr(L, n).restype = d
            getattr(L, n).argtypes = [d, d]
        L.hc_beta_pdf.restype = d
        L.hc_beta_pdf.argtypes = [d, d, d]
        L.hc_davies_pvalue.restype = d
        L.hc_davies_pvalue.argtypes = [c_double_p, C.c_int, d, c_int_p, c_double_p]
        L.hc_davies_pvalue_cached.restype = d
        L.hc_davies_pvalue_cached.argtypes = [c_double_p, C.c_int, d, c_int_p, c_double_p]
        L.hc_davies_pvalue_fast.restype = d
        L.hc_davies_pvalue_fast.argtypes = [c_double_p, C.c_int, d, C.c_int, c_int_p, c_double_p]
        L.hc_liu_pvalue.restype = d
        L.hc_liu_pvalue.argtypes = [c_double_p, C.c_int, d]
        L.hc_sym_eigvals.restype = None
        L.hc_sym_eigvals.argtypes = [c_double_p, C.c_int, c_double_p]
        L.hc_tridiag_eigvals.restype = None
        L.hc_tridiag_eigvals.argtypes = [c_double_p, c_double_p, C.c_int, c_double_p]
        L.hc_qags_builtin.restype = C.c_int
        L.hc_qags_builtin.argtypes = [C.c_int, d, d, d, d, d, C.c_int, c_double_p, c_double_p, c_int_p]
        L.hc_gene.restype = C.c_int
        L.hc_gene.argtypes = [C.c_int, C.c_int64, C.c_int, d, d, d, c_double_p, c_double_p, C.c_int, c_double_p,
                              c_double_p, c_double_p, c_double_p, C.POINTER(Params), C.c_uint,
                              C.POINTER(GeneResult), c_int_p, c_int_p, c_double_p, c_double_p]
        _lib = L
    return _lib


def davies(lam, Q, cached=False, fast=False):
    """Davies p-value through the device algorithm compiled for the host.  fast=False: coefficient sums term by term in
    the reference's order (bit-identical to qfc.c); fast=True: their product form (the engine's default)."""
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    fault = C.c_int(0)
    nt = C.c_double(0)
    if fast:
        p = lib().hc_davies_pvalue_fast(_dp(lam), len(lam), float(Q), 1 if cached else 0, C.byref(fault), C.byref(nt))
        return p, fault.value, nt.value
    fn = lib().hc_davies_pvalue_cached if cached else lib().hc_davies_pvalue
    p = fn(_dp(lam), len(lam), float(Q), C.byref(fault), C.byref(nt))
    return p, fault.value, nt.value


def davies_memo_sweep(lam, Qs):
    """Product-form Davies p-values of many points against ONE shared memo of the searches' coefficient sums (what the
    p-value kernel does across the abscissae of a SKAT-O quadrature); returns p[], (errbd slots, truncation slots) used."""
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    Qs = np.ascontiguousarray(Qs, dtype=np.float64)
    p = np.zeros(len(Qs))
    slots = (C.c_int * 2)()
    L = lib()
    L.hc_davies_memo_sweep.restype = None
    L.hc_davies_memo_sweep.argtypes = [c_double_p, C.c_int, c_double_p, C.c_int, c_double_p, C.POINTER(C.c_int)]
    L.hc_davies_memo_sweep(_dp(lam), len(lam), _dp(Qs), len(Qs), _dp(p), slots)
    return p, (slots[0], slots[1])


def liu(lam, Q):
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    return lib().hc_liu_pvalue(_dp(lam), len(lam), float(Q))


def tridiag_eigvals(d, e):
    d = np.ascontiguousarray(d, dtype=np.float64)
    e = np.ascontiguousarray(np.append(np.asarray(e, dtype=np.float64), 0.0))
    w = np.zeros(len(d))
    lib().hc_tridiag_eigvals(_dp(d), _dp(e), len(d), _dp(w))
    return w


def sym_eigvals(A):
    A = np.asfortranarray(A, dtype=np.float64)
    n = A.shape[0]
    w = np.zeros(n)
    lib().hc_sym_eigvals(_dp(A), n, _dp(w))
    return w


def null_consts(X, res, v, binary):
    """C = X'VX (binary) or X'X (quantitative), its inverse, rss, rsum."""
    X = np.asarray(X, dtype=np.float64)
    if binary:
        Cm = X.T @ (X * v[:, None])
    else:
        Cm = X.T @ X
    return np.ascontiguousarray(Cm), np.ascontiguousarray(np.linalg.inv(Cm)), float(res @ res), float(res.sum())


def suffstats(G, X, res, v, binary):
    """What the MFMA kernel produces: R = G' D [G | X | rr], exact column sum/min/max."""
    G = np.asarray(G, dtype=np.float64)
    if binary:
        Bm = np.column_stack([G, X, res / v])
        R = (G * v[:, None]).T @ Bm
    else:
        Bm = np.column_stack([G, X, res])
        R = G.T @ Bm
    colstat = np.vstack([G.sum(0), G.min(0), G.max(0)])
    return np.ascontiguousarray(R), np.ascontiguousarray(colstat)


def burden_sums(c_cmc, c_zeg, X, res, v, binary):
    out = []
    for c in (c_cmc, c_zeg):
        w = v if binary else np.ones_like(v)
        out.append(np.concatenate([[c @ res, (c * w) @ c, float((c != 0).sum())], (c * w) @ X]))
    return np.ascontiguousarray(np.vstack(out))


def gene(G, af, X, res, v, binary, sigma2, tests=15, params=None, bstats=None):
    G = np.asarray(G, dtype=np.float64)
    N, M = G.shape
    d = X.shape[1]
    Cm, Cinv, rss, rsum = null_consts(X, res, v, binary)
    R, colstat = suffstats(G, X, res, v, binary)
    prm = params or default_params()
    out = GeneResult()
    flip = np.zeros(M, dtype=np.int32)
    kept = np.zeros(M, dtype=np.int32)
    lam = np.zeros(2 * M)
    dbg = np.zeros(64)
    af = np.ascontiguousarray(af, dtype=np.float64)
    bs = _dp(bstats) if bstats is not None else None
    lib().hc_gene(int(binary), N, d, float(sigma2), rss, rsum, _dp(Cm), _dp(Cinv), M, _dp(R), _dp(colstat), bs,
                  _dp(af), C.byref(prm), tests, C.byref(out), flip.ctypes.data_as(c_int_p),
                  kept.ctypes.data_as(c_int_p), _dp(lam), _dp(dbg))
    out.dbg = dbg
    return out, flip, kept, lam


def mvn_band(R, T):
    """(probability, 3.5-sigma error estimate) of P(|Z_i| < T) by the device's deterministic lattice rule."""
    R = np.ascontiguousarray(R, dtype=np.float64)
    err = C.c_double(0.0)
    L = lib()
    L.hc_mvn_band.restype = C.c_double
    L.hc_mvn_band.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_double, C.POINTER(C.c_double)]
    p = L.hc_mvn_band(_dp(R), R.shape[0], float(T), C.byref(err))
    return p, err.value


def mvn_phiinv(p):
    L = lib()
    L.hc_mvn_phiinv.restype = C.c_double
    L.hc_mvn_phiinv.argtypes = [C.c_double]
    return L.hc_mvn_phiinv(float(p))
