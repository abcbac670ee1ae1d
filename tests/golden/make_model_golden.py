#!/usr/bin/env python3
"""Generate tests/golden/model_golden.json: full per-gene results for seeded small problems from an
INDEPENDENT restatement of the reference models, run only in the build container:

  * linear algebra:   numpy / LAPACK (eigvalsh, solve) standing in for Eigen 3.3.9
  * Davies / Liu:     the COMPILED REFERENCE fragment oracle/_ref (MixtureChiSquare.cpp + qfc.c + cdflib.cpp)
  * GSL functions:    scipy.stats / scipy.special (beta pdf, chi-square sf / isf / pdf)
  * integration:      scipy.integrate.quad (QUADPACK QAGS, the routine GSL's qags transcribes) with the reference's
                      epsabs = 1e-25, epsrel = 1.220703e-4, limit = 1000
  * data semantics:   literal numpy restatement of DataConsolidator / collapsers

following regression/Skat.cpp:29-105 (LITERAL N x N P0), regression/SkatO.cpp:60-281, src/Model.h:821-858,
2630-2720, 2787-2860, src/Model.cpp:73-130, src/DataConsolidator.cpp:46-142,217-245.
The fixture stores the seeds/shapes that regenerate the inputs (tests/synth.py) and the expected outputs."""
import json, os, sys
import numpy as np
from scipy import stats, integrate
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc, synth

RHOS = np.minimum(np.arange(11) / 10.0, 0.999)

def flip_poly(G):
    N, M = G.shape
    F = G.copy()
    for j in range(M):
        s = 0.0
        for i in range(N):
            s += G[i, j]
        if not (s <= N):
            F[:, j] = 2 - G[:, j]
    keep = [j for j in range(M) if not np.all(F[:, j] == F[0, j])]
    return F[:, keep]

def weights(af, m, b1, b2, squared):
    w = np.zeros(m)
    for i in range(m):
        f = af[i]
        if f > 0.5: f = 1 - f
        if f > 1e-30:
            w[i] = stats.beta.pdf(f, b1, b2)
            if squared: w[i] *= w[i]
    return w

def davies(lam, Q): return orc.davies(lam, Q, "ref")
def liu(lam, Q): return orc.liu(lam, Q, "ref")

def skat_literal(G, af, X, res, v):
    Gf = flip_poly(G); N, m = Gf.shape
    if m == 0: return None
    w = weights(af, m, 1, 25, True)
    Ks = np.sqrt(w)[:, None] * Gf.T
    Q = float(np.sum((Ks @ res) ** 2))
    if X.shape[1] == 1:
        P0 = -np.outer(v, v) / v.sum(); P0[np.diag_indices(N)] += v
    else:
        XtV = X.T * v
        P0 = -XtV.T @ np.linalg.inv(XtV @ X) @ XtV; P0[np.diag_indices(N)] += v
    ev = np.linalg.eigvalsh(Ks @ P0 @ Ks.T)
    lam = []
    for e in ev[::-1]:
        if e > 1e-30 and len(lam) < min(N, m): lam.append(e)
        else: break
    p = davies(lam, Q)
    if p <= 0 or p == 1: p = liu(lam, Q)
    return dict(Q=Q, p=p, n_poly=m)

def get_eigen(K):
    ev = np.linalg.eigvalsh(K)
    pos = ev[ev > 0]
    if len(pos) == 0: return None
    t = pos.sum() / len(pos) / 100000
    keep = len(ev)
    for e in ev:
        if e < t: keep -= 1
        else: break
    return ev[::-1][:keep]

def moment(la):
    c = [np.sum(la), np.sum(la ** 2), np.sum(la ** 3), np.sum(la ** 4)]
    s1 = c[2] / c[1] / np.sqrt(c[1]); s2 = c[3] / c[1] ** 2
    if s1 * s1 > s2:
        a = 1 / (s1 - np.sqrt(s1 * s1 - s2)); d = s1 * a - a * a; l = a * a - 2 * d
    else:
        l = 1 / s2
    return c[0], 2 * c[1], l

def skato_literal(G, af, X, res, v, binary):
    Gf = flip_poly(G); N, m = Gf.shape
    if m == 0: return None
    w = weights(af, m, 1, 25, False)
    Gw = Gf * w
    if m == 1:
        Q = float((res @ Gw[:, 0]) ** 2)
        if not binary: Q /= (res @ res) / (N - 1)
        Q /= 2
        if not binary:
            W = Gw.T @ Gw - (Gw.T @ X) @ np.linalg.solve(X.T @ X, X.T @ Gw)
        else:
            W = Gw.T @ (Gw * v[:, None]) - (Gw.T @ (X * v[:, None])) @ np.linalg.solve(X.T @ (X * v[:, None]), X.T @ (Gw * v[:, None]))
        lam = get_eigen(W / 2)
        if lam is None: return dict(ok=0, n_poly=m)
        return dict(ok=1, Q=Q, rho=0.0, p=davies(lam, Q), n_poly=m)
    s2 = 1.0 if binary else float(np.linalg.norm(res) ** 2 / (N - 1))
    u = res @ Gw
    Qs = np.array([(u @ (np.full((m, m), r) + (1 - r) * np.eye(m)) @ u) / s2 / 2 for r in RHOS])
    if not binary:
        Z1 = Gw - X @ np.linalg.solve(X.T @ X, X.T @ Gw)
    else:
        sv = np.sqrt(v)
        Z1 = sv[:, None] * Gw - sv[:, None] * (X @ np.linalg.solve(X.T @ (X * v[:, None]), X.T @ (Gw * v[:, None])))
    Z1 = Z1 / np.sqrt(2)
    moms = []
    for r in RHOS:
        L = np.linalg.cholesky(np.full((m, m), r) + (1 - r) * np.eye(m))
        Z2 = Z1 @ L
        lam = get_eigen(Z2.T @ Z2)
        if lam is None: return dict(ok=0, n_poly=m)
        moms.append(moment(lam))
    zbar = Z1.sum(1) / m; z_norm = zbar @ zbar
    zz = zbar @ Z1
    ZMZ = np.outer(zz, zz) / z_norm; ZIMZ = Z1.T @ Z1 - ZMZ
    lam = get_eigen(ZIMZ)
    if lam is None: return dict(ok=0, n_poly=m)
    VarZeta = 4 * np.sum(ZMZ * ZIMZ); MuQ = lam.sum(); VarQ = 2 * np.sum(lam ** 2) + VarZeta
    Df = 12 / (np.sum(lam ** 4) / np.sum(lam ** 2) ** 2 * 12)
    taus = m * m * RHOS * z_norm + (1 - RHOS) * np.sum(zz ** 2) / z_norm
    pvals = np.array([stats.chi2.sf((Qs[i] - mu) / np.sqrt(var) * np.sqrt(2 * df) + df, df) for i, (mu, var, df) in enumerate(moms)])
    minP = pvals.min(); mi = int(np.argmin(pvals))
    qminp = np.array([(stats.chi2.isf(minP, df) - df) / np.sqrt(2 * df) * np.sqrt(var) + mu for (mu, var, df) in moms])
    def integrand(x):
        kappa = np.min((qminp - taus * x) / (1 - RHOS))
        if kappa > lam.sum() * 10000: temp = 0.0
        else:
            Q = (kappa - MuQ) * np.sqrt(VarQ - VarZeta) / np.sqrt(VarQ) + MuQ
            temp = davies(lam, Q)
            if temp <= 0 or temp == 1: temp = liu(lam, Q)
        return (1 - temp) * stats.chi2.pdf(x, 1)
    val, err = integrate.quad(integrand, 0, 40, epsabs=1e-25, epsrel=0.0001220703, limit=1000)
    p = 1 - val
    if p <= 0: p = max(p, minP * 3)
    if p == 0:
        p = pvals[0]
        for q in pvals[1:]:
            if q > 0 and q < p: p = q
    rho = RHOS[mi]
    if rho >= 0.999: rho = 1.0
    return dict(ok=1, Q=float(Qs[mi]), rho=float(rho), p=float(p), n_poly=m)

def burden(G, X, y, binary, which):
    Gf = flip_poly(G); N, m = Gf.shape
    if m == 0: return None
    gi = Gf.astype(np.int64)  # (int)g truncation
    c = (gi > 0).sum(1).astype(float)
    if which == 0: c = (c > 0).astype(float)
    if not binary:
        beta = np.linalg.solve(X.T @ X, X.T @ y); r = y - X @ beta; s2 = (r @ r) / N
        U = c @ r; SS = c @ c - (c @ X) @ np.linalg.solve(X.T @ X, X.T @ c)
        stat = U * U / (SS * s2)
    else:
        rc, b, p, v = orc.fit_logistic(X, y)
        U = (y - p) @ c
        SS = (c * v) @ c - ((c * v) @ X) @ np.linalg.solve(X.T @ (X * v[:, None]), X.T @ (c * v))
        stat = U * U / SS
    return dict(stat=float(stat), p=float(stats.chi2.sf(stat, 1)), nonref=int((c != 0).sum()), csum=float(c.sum()))

def main():
    assert orc.ref() is not None
    cases = []
    specs = [(120, 5, 1, 0, 0), (200, 12, 3, 0, 1), (150, 9, 2, 1, 2), (300, 20, 3, 0, 3), (250, 1, 2, 0, 4),
             (180, 16, 1, 1, 5), (400, 33, 3, 0, 6), (160, 7, 3, 1, 7), (220, 25, 2, 0, 8), (90, 3, 1, 0, 9),
             (350, 40, 3, 0, 10), (140, 10, 1, 1, 11)]
    for (N, M, d, binary, seed) in specs:
        Graw, G, af = synth.make_gene(N, M, seed=seed, missing=0.02 if seed % 2 else 0.0, common=(seed % 3 == 0),
                                      mono=(seed % 4 == 1), maf_hi=-0.9)
        X, y, res, v, s2 = synth.make_null(N, d, binary, seed=100 + seed, G_effect=0.5 * G[:, :2].sum(1))
        rec = dict(N=N, M=M, d=d, binary=binary, seed=seed, missing=0.02 if seed % 2 else 0.0, common=(seed % 3 == 0),
                   mono=(seed % 4 == 1), maf_hi=-0.9)
        rec["skat"] = skat_literal(G, af, X, res, v)
        rec["skato"] = skato_literal(G, af, X, res, v, binary)
        if not (binary and d > 1):
            rec["cmc"] = burden(G, X, y, binary, 0)
            rec["zeggini"] = burden(G, X, y, binary, 1)
        cases.append(rec)
        print(rec["seed"], rec["skat"], rec["skato"])
    json.dump({"source": "numpy/LAPACK + scipy (QUADPACK, cephes/boost) + compiled reference Davies/Liu", "cases": cases},
              open(os.path.join(HERE, "model_golden.json"), "w"), indent=1)

if __name__ == "__main__":
    main()
