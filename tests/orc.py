"""ctypes loader for the CPU oracle (oracle/liboracle.so) and the compiled reference fragment
(oracle/_ref/libref_mixchisq.so).  TEST INFRASTRUCTURE ONLY — nothing under rvtests_amd/ imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


class KernelResult(C.Structure):
    _fields_ = [
        ("fit_ok", C.c_int), ("n_poly", C.c_int), ("Q", C.c_double), ("pvalue", C.c_double),
        ("rho", C.c_double), ("n_lambda", C.c_int), ("lambda_", C.c_double * 512),
        ("Qs", C.c_double * 11), ("pvals", C.c_double * 11), ("taus", C.c_double * 11),
        ("qminp", C.c_double * 11), ("muQ", C.c_double), ("varQ", C.c_double), ("varZeta", C.c_double),
        ("df", C.c_double), ("minP", C.c_double), ("qags_status", C.c_int), ("qags_neval", C.c_int),
    ]


class BurdenResult(C.Structure):
    _fields_ = [("fit_ok", C.c_int), ("n_poly", C.c_int), ("nonref_site", C.c_int), ("U", C.c_double),
                ("V", C.c_double), ("stat", C.c_double), ("pvalue", C.c_double)]


class PermResult(C.Structure):
    _fields_ = [("num_perm", C.c_int), ("actual_perm", C.c_int), ("num_x", C.c_int), ("num_equal", C.c_int),
                ("threshold", C.c_double), ("pvalue", C.c_double)]


def build(native=False):
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])
    if os.path.isdir("/root/reference/regression"):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


_lib = None
_ref = None


def lib(native=False):
    global _lib
    if _lib is None:
        # ORC_LIBRARY: another build of the same sources (oracle/liboracle_native.so: -O3 -march=native; bench.py times it
        # beside the reference-flags build)
        path = os.environ.get("ORC_LIBRARY") or os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        d = C.c_double
        for name, args in [
            ("orc_gamma_inc_P", [d, d]), ("orc_gamma_inc_Q", [d, d]), ("orc_chisq_P", [d, d]),
            ("orc_chisq_Q", [d, d]), ("orc_chisq_Qinv", [d, d]), ("orc_chisq_pdf", [d, d]),
            ("orc_beta_pdf", [d, d, d]), ("orc_gamma_pdf", [d, d, d]),
            ("orc_gamma_cdf_P", [d, d, d]), ("orc_gamma_cdf_Q", [d, d, d]), ("orc_gamma_cdf_Qinv", [d, d, d]),
        ]:
            f = getattr(L, name)
            f.restype = d
            f.argtypes = args
        L.orc_qf.restype = d
        L.orc_qf.argtypes = [c_double_p, c_double_p, c_int_p, C.c_int, d, d, C.c_int, d, c_double_p, c_int_p]
        L.orc_davies_pvalue.restype = d
        L.orc_davies_pvalue.argtypes = [c_double_p, C.c_int, d, c_int_p]
        L.orc_liu_pvalue.restype = d
        L.orc_liu_pvalue.argtypes = [c_double_p, C.c_int, d]
        L.orc_cumchn.restype = None
        L.orc_cumchn.argtypes = [d, d, d, c_double_p, c_double_p]
        L.orc_qags_builtin.restype = C.c_int
        L.orc_qags_builtin.argtypes = [C.c_int, d, d, d, d, d, C.c_int, c_double_p, c_double_p, c_int_p]
        L.orc_sym_eigvals.restype = None
        L.orc_sym_eigvals.argtypes = [c_double_p, C.c_int, c_double_p]
        L.orc_fit_linear.restype = C.c_int
        L.orc_fit_linear.argtypes = [c_double_p, c_double_p, C.c_int64, C.c_int, c_double_p, c_double_p,
                                     c_double_p, c_double_p]
        L.orc_fit_logistic.restype = C.c_int
        L.orc_fit_logistic.argtypes = [c_double_p, c_double_p, C.c_int64, C.c_int, C.c_int, c_double_p,
                                       c_double_p, c_double_p]
        L.orc_impute_mean.restype = None
        L.orc_impute_mean.argtypes = [c_double_p, C.c_int64, C.c_int]
        L.orc_counter_af.restype = None
        L.orc_counter_af.argtypes = [c_double_p, C.c_int64, C.c_int, c_double_p]
        L.orc_flip_poly.restype = C.c_int
        L.orc_flip_poly.argtypes = [c_double_p, C.c_int64, C.c_int, c_double_p, c_int_p, c_int_p]
        L.orc_collapse.restype = None
        L.orc_collapse.argtypes = [c_double_p, C.c_int64, C.c_int, C.c_int, c_double_p]
        kargs = [c_double_p, c_double_p, C.c_int64, C.c_int, c_double_p, C.c_int, c_double_p, c_double_p,
                 C.c_int, d, d, C.POINTER(KernelResult)]
        L.orc_skat.restype = C.c_int
        L.orc_skat.argtypes = kargs
        L.orc_skato.restype = C.c_int
        L.orc_skato.argtypes = kargs
        L.orc_skat_literal.restype = C.c_int
        L.orc_skat_literal.argtypes = kargs[:-1] + [C.c_int, C.POINTER(KernelResult)]
        L.orc_burden.restype = C.c_int
        L.orc_burden.argtypes = [c_double_p, C.c_int64, C.c_int, c_double_p, C.c_int, c_double_p, C.c_int,
                                 C.c_int, C.POINTER(BurdenResult)]
        L.orc_rand_seed.restype = None
        L.orc_rand_seed.argtypes = [C.c_uint]
        L.orc_rand.restype = C.c_int
        L.orc_skat_permute.restype = C.c_int
        L.orc_skat_permute.argtypes = [c_double_p, c_double_p, C.c_int64, C.c_int, c_double_p, d, d, d, C.c_int,
                                       d, C.c_int, C.POINTER(PermResult)]
        _lib = L
    return _lib


def ref():
    """The compiled reference fragment, or None when it has not been built / is absent."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libref_mixchisq.so")
        if not os.path.exists(path):
            if os.path.isdir("/root/reference/regression"):
                subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])
            else:
                return None
        R = C.CDLL(path)
        d = C.c_double
        R.ref_davies_pvalue.restype = d
        R.ref_davies_pvalue.argtypes = [c_double_p, C.c_int, d]
        R.ref_liu_pvalue.restype = d
        R.ref_liu_pvalue.argtypes = [c_double_p, C.c_int, d]
        R.ref_qf.restype = d
        R.ref_qf.argtypes = [c_double_p, c_double_p, c_int_p, C.c_int, d, d, C.c_int, d, c_double_p, c_int_p]
        R.ref_cumchn.restype = None
        R.ref_cumchn.argtypes = [d, d, d, c_double_p, c_double_p]
        R.ref_cumchi.restype = None
        R.ref_cumchi.argtypes = [d, d, c_double_p, c_double_p]
        R.ref_gamma_inc.restype = None
        R.ref_gamma_inc.argtypes = [d, d, c_double_p, c_double_p]
        _ref = R
    return _ref


_ref_counter = None


def ref_counter():
    """The reference's GenotypeCounter (+ SNPHWE) and RingMemoryPool compiled where they lie (oracle/_ref/libref_counter.so),
    or None when it has not been built / is absent."""
    global _ref_counter
    if _ref_counter is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libref_counter.so")
        if not os.path.exists(path):
            if os.path.isdir("/root/reference/src"):
                subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])
            else:
                return None
        R = C.CDLL(path)
        R.ref_counter.restype = None
        R.ref_counter.argtypes = [c_double_p, C.c_longlong, c_double_p]
        R.ref_ring_new.restype = C.c_void_p
        R.ref_ring_new.argtypes = [C.c_int, C.c_int]
        R.ref_ring_delete.restype = None
        R.ref_ring_delete.argtypes = [C.c_void_p]
        R.ref_ring_allocate.restype = C.c_int
        R.ref_ring_allocate.argtypes = [C.c_void_p]
        R.ref_ring_deallocate.restype = None
        R.ref_ring_deallocate.argtypes = [C.c_void_p, C.c_int]
        R.ref_ring_chunk.restype = C.POINTER(C.c_float)
        R.ref_ring_chunk.argtypes = [C.c_void_p, C.c_int]
        R.ref_ring_size.restype = C.c_longlong
        R.ref_ring_size.argtypes = [C.c_void_p]
        R.ref_ring_capacity.restype = C.c_longlong
        R.ref_ring_capacity.argtypes = [C.c_void_p]
        _ref_counter = R
    return _ref_counter


# ------------------------------------------------------------------ convenience wrappers
def F(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def davies(lam, Q, which="orc"):
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    if which == "ref":
        return ref().ref_davies_pvalue(_dp(lam), len(lam), float(Q))
    fault = C.c_int(0)
    return lib().orc_davies_pvalue(_dp(lam), len(lam), float(Q), C.byref(fault))


def liu(lam, Q, which="orc"):
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    if which == "ref":
        return ref().ref_liu_pvalue(_dp(lam), len(lam), float(Q))
    return lib().orc_liu_pvalue(_dp(lam), len(lam), float(Q))


def qf(lam, c, which="orc", lim=10000, acc=1e-6, sigma=0.0):
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    r = len(lam)
    nc = np.zeros(r)
    n = np.ones(r, dtype=np.int32)
    trace = np.zeros(7)
    fault = C.c_int(0)
    fn = ref().ref_qf if which == "ref" else lib().orc_qf
    val = fn(_dp(lam), _dp(nc), _ip(n), r, sigma, float(c), lim, acc, _dp(trace), C.byref(fault))
    return val, fault.value, trace


def sym_eigvals(A):
    A = F(A)
    n = A.shape[0]
    w = np.zeros(n)
    lib().orc_sym_eigvals(_dp(A), n, _dp(w))
    return w


def fit_linear(X, y):
    X = F(X)
    y = np.ascontiguousarray(y, dtype=np.float64)
    N, d = X.shape
    beta = np.zeros(d)
    pred = np.zeros(N)
    resid = np.zeros(N)
    s2 = C.c_double(0)
    rc = lib().orc_fit_linear(_dp(X), _dp(y), N, d, _dp(beta), _dp(pred), _dp(resid), C.byref(s2))
    return rc, beta, pred, resid, s2.value


def fit_logistic(X, y, rounds=100):
    X = F(X)
    y = np.ascontiguousarray(y, dtype=np.float64)
    N, d = X.shape
    beta = np.zeros(d)
    p = np.zeros(N)
    v = np.zeros(N)
    rc = lib().orc_fit_logistic(_dp(X), _dp(y), N, d, rounds, _dp(beta), _dp(p), _dp(v))
    return rc, beta, p, v


def impute_mean(Graw):
    G = F(Graw).copy(order="F")
    N, M = G.shape
    lib().orc_impute_mean(_dp(G), N, M)
    return G


def counter_af(Graw):
    G = F(Graw)
    N, M = G.shape
    af = np.zeros(M)
    lib().orc_counter_af(_dp(G), N, M, _dp(af))
    return af


def flip_poly(G):
    G = F(G)
    N, M = G.shape
    out = np.zeros((N, M), order="F")
    fl = np.zeros(M, dtype=np.int32)
    kp = np.zeros(M, dtype=np.int32)
    m = lib().orc_flip_poly(_dp(G), N, M, _dp(out), _ip(fl), _ip(kp))
    return np.asfortranarray(out.reshape(-1, order="F")[: N * m].reshape((N, m), order="F")), fl, kp


def collapse(Gf, which):
    Gf = F(Gf)
    N, M = Gf.shape
    out = np.zeros(N)
    lib().orc_collapse(_dp(Gf), N, M, which, _dp(out))
    return out


def _kernel(fn, G, af, X, res, v, binary, b1, b2, extra=None):
    G = F(G)
    X = F(X)
    N, M = G.shape
    d = X.shape[1]
    af = np.ascontiguousarray(af, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = KernelResult()
    args = [_dp(G), _dp(af), N, M, _dp(X), d, _dp(res), _dp(v), int(binary), b1, b2]
    if extra is not None:
        args.append(extra)
    rc = fn(*args, C.byref(out))
    return rc, out


def skat(G, af, X, res, v, binary=0, b1=1.0, b2=25.0):
    return _kernel(lib().orc_skat, G, af, X, res, v, binary, b1, b2)


def skat_literal(G, af, X, res, v, binary=0, b1=1.0, b2=25.0, use_float=0):
    return _kernel(lib().orc_skat_literal, G, af, X, res, v, binary, b1, b2, extra=use_float)


def skato(G, af, X, res, v, binary=0, b1=1.0, b2=25.0):
    return _kernel(lib().orc_skato, G, af, X, res, v, binary, b1, b2)


def burden(G, X, y, binary, which):
    G = F(G)
    X = F(X)
    N, M = G.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = BurdenResult()
    rc = lib().orc_burden(_dp(G), N, M, _dp(X), X.shape[1], _dp(y), int(binary), int(which), C.byref(out))
    return rc, out


def metascore(G, X, y, binary):
    """MetaScoreTest (unrelated samples) restatement: dict of ok, U, V, effect, se, p (V entries) + beta, covb, sigma2."""
    G = F(G)
    X = F(X)
    N, V = G.shape
    d = X.shape[1]
    y = np.ascontiguousarray(y, dtype=np.float64)
    ok = np.zeros(V, dtype=np.int32)
    arr = [np.zeros(V) for _ in range(5)]
    beta, covb = np.zeros(d), np.zeros(d)
    s2 = C.c_double(0)
    L = lib()
    L.orc_metascore.restype = C.c_int
    rc = L.orc_metascore(_dp(G), C.c_int64(N), V, _dp(X), d, _dp(y), int(binary),
                         ok.ctypes.data_as(C.POINTER(C.c_int)), *[_dp(a) for a in arr], _dp(beta), _dp(covb),
                         C.byref(s2))
    return rc, dict(ok=ok, U=arr[0], V=arr[1], effect=arr[2], se=arr[3], p=arr[4], beta=beta, covb=covb,
                    sigma2=s2.value)


def metacov(G, chrom, pos, X, y, binary, window, use_float=False):
    """MetaCovTest (unrelated samples) restatement: returns rc, kept[V], cov[V, V] (cov[h, j] for j >= h in h's
    row, NaN elsewhere; unscaled), row_end[V], xz[V, d], zz[d, d]."""
    G = F(G)
    X = F(X)
    N, V = G.shape
    d = X.shape[1]
    y = np.ascontiguousarray(y, dtype=np.float64)
    chrom = np.ascontiguousarray(chrom, dtype=np.int32)
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    kept = np.zeros(V, dtype=np.int32)
    row_end = np.zeros(V, dtype=np.int32)
    cov = np.zeros((V, V), order="F")
    xz = np.zeros((V, d))
    zz = np.zeros((d, d))
    L = lib()
    L.orc_metacov.restype = C.c_int
    rc = L.orc_metacov(_dp(G), C.c_int64(N), V, _ip(chrom), _ip(pos), _dp(X), _dp(y), d, int(binary), int(window),
                       int(use_float), _ip(kept), _dp(cov), _ip(row_end), _dp(xz), _dp(zz))
    return rc, kept, cov, row_end, xz, zz


class FamNull(C.Structure):
    _fields_ = [("ok", C.c_int), ("max_index", C.c_int), ("brent_evals", C.c_int), ("delta", C.c_double),
                ("sigma2", C.c_double), ("beta", C.c_double * 16)]


def fastlmm_null(X, y, U, S, use_float=False):
    X = F(X)
    U = F(U)
    N, d = X.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    S = np.ascontiguousarray(S, dtype=np.float64)
    out = FamNull()
    rc = lib().orc_fastlmm_null(_dp(X), _dp(y), C.c_int64(N), d, _dp(U), _dp(S), int(use_float), C.byref(out))
    return rc, out


def famskat(G, X, y, U, S, nul, use_float=False):
    G = F(G)
    X = F(X)
    U = F(U)
    N, M = G.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    S = np.ascontiguousarray(S, dtype=np.float64)
    out = KernelResult()
    rc = lib().orc_famskat(_dp(G), C.c_int64(N), M, _dp(X), _dp(y), X.shape[1], _dp(U), _dp(S), C.byref(nul),
                           int(use_float), C.byref(out))
    return rc, out


def rand_seed(seed=1):
    lib().orc_rand_seed(int(seed))


def skat_permute(G, af, res, obs, n_perm, alpha, b1=1.0, b2=25.0, use_float=False):
    """SkatTest's permutation loop on the oracle's glibc rand() emulation (continues the oracle's global stream)."""
    G = F(G)
    N, M = G.shape
    af = np.ascontiguousarray(af, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.float64)
    out = PermResult()
    rc = lib().orc_skat_permute(_dp(G), _dp(af), C.c_int64(N), M, _dp(res), float(b1), float(b2), float(obs),
                                int(n_perm), float(alpha), int(use_float), C.byref(out))
    return rc, out


def metacov_fam(G, chrom, pos, X, U, S, nul, window, use_float=False):
    G = F(G)
    X = F(X)
    U = F(U)
    N, V = G.shape
    d = X.shape[1]
    S = np.ascontiguousarray(S, dtype=np.float64)
    chrom = np.ascontiguousarray(chrom, dtype=np.int32)
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    kept = np.zeros(V, dtype=np.int32)
    row_end = np.zeros(V, dtype=np.int32)
    cov = np.zeros((V, V), order="F")
    xz = np.zeros((V, d))
    zz = np.zeros((d, d))
    L = lib()
    L.orc_metacov_fam.restype = C.c_int
    rc = L.orc_metacov_fam(_dp(G), C.c_int64(N), V, _ip(chrom), _ip(pos), _dp(X), d, _dp(U), _dp(S), C.byref(nul),
                           int(window), int(use_float), _ip(kept), _dp(cov), _ip(row_end), _dp(xz), _dp(zz))
    return rc, kept, cov, row_end, xz, zz


def obtain_b(alpha):
    L = lib()
    L.orc_obtain_b.restype = C.c_double
    L.orc_obtain_b.argtypes = [C.c_double]
    return L.orc_obtain_b(float(alpha))


def metacov_fam_binary(G, chrom, pos, X, y, U, S, nul, window, use_float=False):
    G = F(G)
    X = F(X)
    U = F(U)
    N, V = G.shape
    d = X.shape[1]
    S = np.ascontiguousarray(S, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    chrom = np.ascontiguousarray(chrom, dtype=np.int32)
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    kept = np.zeros(V, dtype=np.int32)
    row_end = np.zeros(V, dtype=np.int32)
    cov = np.zeros((V, V), order="F")
    xz = np.zeros((V, d))
    zz = np.zeros((d, d))
    L = lib()
    L.orc_metacov_fam_binary.restype = C.c_int
    rc = L.orc_metacov_fam_binary(_dp(G), C.c_int64(N), V, _ip(chrom), _ip(pos), _dp(X), d, _dp(y), _dp(U), _dp(S),
                                  C.byref(nul), int(window), int(use_float), _ip(kept), _dp(cov), _ip(row_end), _dp(xz),
                                  _dp(zz))
    return rc, kept, cov, row_end, xz, zz


class FamBurdenResult(C.Structure):
    _fields_ = [("fit_ok", C.c_int), ("num_site", C.c_int), ("af", C.c_double), ("U", C.c_double), ("V", C.c_double),
                ("stat", C.c_double), ("pvalue", C.c_double)]


def fam_burden(G, X, y, U, S, nul, which, use_float=False):
    G = F(G)
    X = F(X)
    U = F(U)
    N, M = G.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    S = np.ascontiguousarray(S, dtype=np.float64)
    out = FamBurdenResult()
    L = lib()
    L.orc_fam_burden.restype = C.c_int
    rc = L.orc_fam_burden(_dp(G), C.c_int64(N), M, _dp(X), _dp(y), X.shape[1], _dp(U), _dp(S), C.byref(nul),
                          int(which), int(use_float), C.byref(out))
    return rc, out


def fastlmm_covb(X, U, S, delta, use_float=False):
    """FastLMM::GetNullCovB restated: (ux' diag(|S| + delta) ux)^-1, d x d."""
    X = F(X)
    U = F(U)
    N, d = X.shape
    S = np.ascontiguousarray(S, dtype=np.float64)
    out = np.zeros((d, d))
    L = lib()
    L.orc_fastlmm_covb.restype = C.c_int
    rc = L.orc_fastlmm_covb(_dp(X), C.c_int64(N), d, _dp(U), _dp(S), C.c_double(delta), int(use_float), _dp(out))
    return rc, out


# ---- VCF genotype text rules (oracle/orc_vcf.cpp; reference fragment oracle/_ref/libref_vcf.so) -----------------------
_ref_vcf = None


def ref_vcf():
    """The reference's own libVcf value / individual parsers, or None where they are not built."""
    global _ref_vcf
    if _ref_vcf is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libref_vcf.so")
        if not os.path.exists(path):
            if os.path.isdir("/root/reference/libVcf"):
                subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])
            else:
                return None
        R = C.CDLL(path)
        R.ref_vcf_column_genotype.restype = C.c_int
        R.ref_vcf_column_genotype.argtypes = [C.c_char_p, C.c_int, C.c_int]
        R.ref_vcf_column_alt.restype = C.c_int
        R.ref_vcf_column_alt.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        R.ref_vcf_column_male02.restype = C.c_int
        R.ref_vcf_column_male02.argtypes = [C.c_char_p, C.c_int, C.c_int]
        R.ref_vcf_column_male_alt.restype = C.c_int
        R.ref_vcf_column_male_alt.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        R.ref_vcf_column_int.restype = C.c_int
        R.ref_vcf_column_int.argtypes = [C.c_char_p, C.c_int, C.c_int]
        _ref_vcf = R
    return _ref_vcf


def vcf_column_genotype(col, gt_idx):
    L = lib()
    L.orc_vcf_column_genotype.restype = C.c_int
    L.orc_vcf_column_genotype.argtypes = [C.c_char_p, C.c_int64, C.c_int]
    return L.orc_vcf_column_genotype(col, len(col), gt_idx)


def vcf_column_male02(col, gt_idx):
    L = lib()
    L.orc_vcf_column_male02.restype = C.c_int
    L.orc_vcf_column_male02.argtypes = [C.c_char_p, C.c_int64, C.c_int]
    return L.orc_vcf_column_male02(col, len(col), gt_idx)


def vcf_column_male_alt(col, gt_idx, alt):
    L = lib()
    L.orc_vcf_column_male_alt.restype = C.c_int
    L.orc_vcf_column_male_alt.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int]
    return L.orc_vcf_column_male_alt(col, len(col), gt_idx, alt)


def vcf_decode_record_sex(text, row_of_sample, n_rows, gt_idx, gd_idx=-1, gq_idx=-1, filters=(0, 0, 0, 0), alt=0, hemi=0,
                          sex=None):
    """vcf_decode_record with the multi-allelic allele, the record's hemizygous flag and the file samples' PLINK sex."""
    L = lib()
    L.orc_vcf_decode_record_sex.restype = C.c_int
    L.orc_vcf_decode_record_sex.argtypes = [C.c_char_p, C.c_int64, C.c_int, c_int_p, C.c_int, C.c_int, C.c_int, c_int_p,
                                            C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int8)]
    rows = np.ascontiguousarray(row_of_sample, dtype=np.int32)
    flt = np.ascontiguousarray(filters, dtype=np.int32)
    sx = None if sex is None else np.ascontiguousarray(sex, dtype=np.int8)
    out = np.full(n_rows, -9, dtype=np.int8)
    n = L.orc_vcf_decode_record_sex(text, len(text), len(rows), rows.ctypes.data_as(c_int_p), gt_idx, gd_idx, gq_idx,
                                    flt.ctypes.data_as(c_int_p), alt, hemi, None if sx is None else sx.ctypes.data,
                                    out.ctypes.data_as(C.POINTER(C.c_int8)))
    return out, n


def vcf_decode_record_dosage_sex(text, row_of_sample, n_rows, tag_idx, gd_idx=-1, gq_idx=-1, filters=(0, 0, 0, 0), hemi=0,
                                 sex=None):
    L = lib()
    L.orc_vcf_decode_record_dosage_sex.restype = C.c_int
    L.orc_vcf_decode_record_dosage_sex.argtypes = [C.c_char_p, C.c_int64, C.c_int, c_int_p, C.c_int, C.c_int, C.c_int,
                                                   c_int_p, C.c_int, C.c_void_p, c_double_p]
    rows = np.ascontiguousarray(row_of_sample, dtype=np.int32)
    flt = np.ascontiguousarray(filters, dtype=np.int32)
    sx = None if sex is None else np.ascontiguousarray(sex, dtype=np.int8)
    out = np.full(n_rows, -9.0, dtype=np.float64)
    n = L.orc_vcf_decode_record_dosage_sex(text, len(text), len(rows), rows.ctypes.data_as(c_int_p), tag_idx, gd_idx,
                                           gq_idx, flt.ctypes.data_as(c_int_p), hemi,
                                           None if sx is None else sx.ctypes.data, out.ctypes.data_as(c_double_p))
    return out, n


def vcf_format_index(fmt, key):
    L = lib()
    L.orc_vcf_format_index.restype = C.c_int
    L.orc_vcf_format_index.argtypes = [C.c_char_p, C.c_int64, C.c_char_p]
    return L.orc_vcf_format_index(fmt, len(fmt), key)


def vcf_decode_record(text, row_of_sample, n_rows, gt_idx, gd_idx=-1, gq_idx=-1, filters=(0, 0, 0, 0)):
    """(codes[n_rows] int8 with -9 where no column addressed the row, number of columns found)."""
    L = lib()
    L.orc_vcf_decode_record.restype = C.c_int
    L.orc_vcf_decode_record.argtypes = [C.c_char_p, C.c_int64, C.c_int, c_int_p, C.c_int, C.c_int, C.c_int, c_int_p,
                                        C.POINTER(C.c_int8)]
    rows = np.ascontiguousarray(row_of_sample, dtype=np.int32)
    flt = np.ascontiguousarray(filters, dtype=np.int32)
    out = np.full(n_rows, -9, dtype=np.int8)
    n = L.orc_vcf_decode_record(text, len(text), len(rows), rows.ctypes.data_as(c_int_p), gt_idx, gd_idx, gq_idx,
                                flt.ctypes.data_as(c_int_p), out.ctypes.data_as(C.POINTER(C.c_int8)))
    return out, n


# ---- AnalyticVT (oracle/orc_vt.cpp; reference fragment oracle/_ref/libref_mvt.so = the reference's mvt.f) -----------------
class VtResult(C.Structure):
    _fields_ = [("fit_ok", C.c_int), ("n_poly", C.c_int), ("opt_num", C.c_int), ("n_cutoff", C.c_int),
                ("min_maf", C.c_double), ("max_maf", C.c_double), ("opt_maf", C.c_double), ("U", C.c_double),
                ("V", C.c_double), ("stat", C.c_double), ("pvalue", C.c_double), ("p_err", C.c_double)]


def analytic_vt(G, af, X, y, mvn_points=4096):
    """(rc, VtResult, correlation matrix of the threshold statistics)."""
    G = F(G)
    X = F(X)
    N, M = G.shape
    d = X.shape[1]
    af = np.ascontiguousarray(af, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = VtResult()
    cor = np.zeros((M, M))
    L = lib()
    L.orc_analytic_vt.restype = C.c_int
    rc = L.orc_analytic_vt(_dp(G), _dp(af), _dp(X), _dp(y), C.c_int64(N), M, d, C.c_longlong(mvn_points), C.byref(out),
                           _dp(cor))
    K = out.n_cutoff
    return rc, out, cor.ravel()[:K * K].reshape(K, K).copy()


def mvn_band(R, T, points=4096):
    R = np.ascontiguousarray(R, dtype=np.float64)
    err = C.c_double(0.0)
    L = lib()
    L.orc_mvn_band.restype = C.c_double
    L.orc_mvn_band.argtypes = [c_double_p, C.c_int, C.c_double, C.c_longlong, c_double_p]
    p = L.orc_mvn_band(_dp(R), R.shape[0], float(T), points, C.byref(err))
    return p, err.value


_ref_mvt = None


def ref_mvt():
    """The reference's own MVTDST (mvt.f built with flang), or None where it is not built."""
    global _ref_mvt
    if _ref_mvt is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libref_mvt.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_mvn_band.restype = C.c_int
        R.ref_mvn_band.argtypes = [C.c_int, C.c_double, c_double_p, C.c_uint, c_double_p, c_double_p]
        _ref_mvt = R
    return _ref_mvt


def ref_mvn_band(R, T, seed=1):
    """(inform, probability, error estimate) from the reference's MVTDST with MvtNorm's settings."""
    lib_ = ref_mvt()
    R = np.asarray(R, dtype=np.float64)
    n = R.shape[0]
    packed = np.array([R[i, j] for i in range(1, n) for j in range(i)], dtype=np.float64)
    if packed.size == 0:
        packed = np.zeros(1)
    prob, err = C.c_double(0.0), C.c_double(0.0)
    inform = lib_.ref_mvn_band(n, float(T), _dp(packed), int(seed), C.byref(prob), C.byref(err))
    return inform, prob.value, err.value


# ---- KBAC (oracle/orc_kbac.cpp) ---------------------------------------------------------------------------------------------
def hypergeometric_P(k, n1, n2, t):
    L = lib()
    L.orc_hypergeometric_P.restype = C.c_double
    L.orc_hypergeometric_P.argtypes = [C.c_uint] * 4
    return L.orc_hypergeometric_P(int(k), int(n1), int(n2), int(t))


def kbac(Gf, y, maf, nperm, alpha, seed=None):
    """Gf: N x M flipped / polymorphic / imputed genotype.  Returns (pvalue, observed statistic, patterns, permutations
    done, next rand()).  seed: srand(seed) first (None: continue the stream)."""
    L = lib()
    Gf = np.ascontiguousarray(Gf, dtype=np.float64)          # people-major
    N, M = Gf.shape
    y = np.ascontiguousarray(y, dtype=np.float64)
    maf = np.ascontiguousarray(maf, dtype=np.float64)
    if seed is not None:
        L.orc_rand_seed(int(seed))
    p, obs = C.c_double(0.0), C.c_double(0.0)
    npat, done = C.c_int(0), C.c_int(0)
    L.orc_kbac.restype = C.c_int
    L.orc_kbac(_dp(Gf), _dp(y), _dp(maf), N, M, int(nperm), C.c_double(alpha), C.byref(p), C.byref(obs), C.byref(npat),
               C.byref(done))
    return p.value, obs.value, npat.value, done.value


def fam_analytic_vt(G, X, y, U, S, nul, mvn_points=2048):
    """FamAnalyticVT restated literally (N x N scaledK).  nul: FamNull (delta, sigma2, beta)."""
    G = F(G)
    X = F(X)
    U = F(U)
    N, M = G.shape
    d = X.shape[1]
    S = np.ascontiguousarray(S, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    beta = np.array([nul.beta[k] for k in range(d)], dtype=np.float64)
    out = VtResult()
    cor = np.zeros((M, M))
    L = lib()
    L.orc_fam_analytic_vt.restype = C.c_int
    rc = L.orc_fam_analytic_vt(_dp(G), _dp(X), _dp(y), C.c_int64(N), M, d, _dp(U), _dp(S), C.c_double(nul.delta),
                               C.c_double(nul.sigma2), _dp(beta), C.c_longlong(mvn_points), C.byref(out), _dp(cor))
    K = out.n_cutoff
    return rc, out, cor.ravel()[:K * K].reshape(K, K).copy()


def vcf_decode_record_dosage(text, row_of_sample, n_rows, tag_idx, gd_idx=-1, gq_idx=-1, filters=(0, 0, 0, 0)):
    L = lib()
    L.orc_vcf_decode_record_dosage.restype = C.c_int
    L.orc_vcf_decode_record_dosage.argtypes = [C.c_char_p, C.c_int64, C.c_int, c_int_p, C.c_int, C.c_int, C.c_int,
                                               c_int_p, c_double_p]
    rows = np.ascontiguousarray(row_of_sample, dtype=np.int32)
    flt = np.ascontiguousarray(filters, dtype=np.int32)
    out = np.full(n_rows, -9.0)
    n = L.orc_vcf_decode_record_dosage(text, len(text), len(rows), rows.ctypes.data_as(c_int_p), tag_idx, gd_idx, gq_idx,
                                       flt.ctypes.data_as(c_int_p), _dp(out))
    return out, n


def vcf_column_alt(col, gt_idx, alt):
    L = lib()
    L.orc_vcf_column_alt.restype = C.c_int
    L.orc_vcf_column_alt.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int]
    return L.orc_vcf_column_alt(col, len(col), gt_idx, alt)


# ---- BGEN genotype-probability blocks (oracle/orc_bgen.cpp) -------------------------------------------------------------
def bgen_decode(block, layout, N):
    """-> (info dict, missing[N], ploidy[N], index[N+1], prob float32[...]) as BGenFile::parseLayout1/2 leave them."""
    L = lib()
    L.orc_bgen_decode.restype = C.c_int64
    L.orc_bgen_decode.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int64, c_int_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int64]
    missing = np.zeros(N, dtype=np.uint8)
    ploidy = np.zeros(N, dtype=np.uint8)
    index = np.zeros(N + 1, dtype=np.int64)
    info = np.zeros(3, dtype=np.int32)
    cap = 3 * N + 16
    while True:
        prob = np.zeros(cap, dtype=np.float32)
        n = L.orc_bgen_decode(bytes(block), len(block), layout, N, info.ctypes.data_as(c_int_p), missing.ctypes.data,
                              ploidy.ctypes.data, index.ctypes.data, prob.ctypes.data, cap)
        if n != -2:
            break
        cap *= 4
    if n < 0:
        raise ValueError("orc_bgen_decode: %d" % n)
    return {"phased": int(info[0]), "bits": int(info[1]), "K": int(info[2])}, missing, ploidy, index, prob[:n]


def bgen_block_genotypes(block, layout, N):
    """BGenGenotypeExtractor::getGenotype for every file sample of one uncompressed block (-9 = missing)."""
    L = lib()
    L.orc_bgen_block_genotypes.restype = C.c_int
    L.orc_bgen_block_genotypes.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int64, c_double_p]
    out = np.zeros(N, dtype=np.float64)
    rc = L.orc_bgen_block_genotypes(bytes(block), len(block), layout, N, out.ctypes.data_as(c_double_p))
    if rc:
        raise ValueError("orc_bgen_block_genotypes: %d" % rc)
    return out
