#!/usr/bin/env python3
"""How unevenly the abscissae of a SKAT-O quadrature panel load qf()'s searches (CPU; profiling build of the host harness).

The p-value kernel evaluates the 21 / 42 abscissae of a QAGS step on the lanes of one wave; a wave pays for its WORST lane.
This tool runs the serial form of the stage (rvtests_amd/csrc/hostcheck.cpp built with -DRVT_DV_PROFILE) on synthetic genes
and prints, per gene, the errbd + truncation evaluations summed over the panels as the MEAN lane sees them and as the MAX
lane does.

usage: python tools/davies_divergence.py [--genes 8] [--n 4000] [--m 50]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=8)
    ap.add_argument("--n", type=int, default=4000)
    ap.add_argument("--m", type=int, default=50)
    a = ap.parse_args()
    out = "/tmp/librvt_hostcheck_prof.so"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-DRVT_DV_PROFILE", "-o", out,
                           os.path.join(hc.CSRC, "hostcheck.cpp"), "-lm", "-lpthread"])
    os.environ["RVT_HOSTCHECK_LIB"] = out
    hc._lib = None
    lib = hc.lib()
    lib.hc_dv_log.restype = C.c_int
    lib.hc_dv_log.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    rng = np.random.default_rng(1)
    N, M = a.n, a.m
    X = np.column_stack([np.ones(N), rng.normal(size=(N, 2))])
    for g in range(a.genes):
        maf = rng.uniform(0.002, 0.05, size=M)
        G = (rng.random((N, M)) < maf).astype(float) + (rng.random((N, M)) < maf).astype(float)
        y = X @ np.array([0.2, 0.5, -0.3]) + rng.normal(size=N) + (0.4 * G[:, :5].sum(1) if g % 2 else 0.0)
        beta = np.linalg.lstsq(X, y, rcond=None)[0]
        res = y - X @ beta
        sigma2 = float(res @ res / (N - X.shape[1]))
        af = G.mean(0) / 2
        hc.gene(G, af, X, res, np.ones(N), 0, sigma2)
        keys = (C.c_longlong * 4)()
        lib.hc_dv_keys(keys)
        print("   evaluation points: errbd %d distinct of %d, truncation %d distinct of %d" % (keys[0], keys[1], keys[2], keys[3]))
        buf = (C.c_longlong * (5 * 20000))()
        n = lib.hc_dv_log(buf, len(buf))
        cum = np.array(buf[:5 * n], dtype=np.int64).reshape(n, 5)
        per = np.diff(np.vstack([np.zeros((1, 5), dtype=np.int64), cum]), axis=0)
        work = per[:, 0] + per[:, 1]                      # errbd + truncation evaluations of the abscissa
        panels = [work[:21]] + [work[i:i + 42] for i in range(21, n, 42)]
        mean_sum = sum(p.mean() for p in panels)
        max_sum = sum(p.max() for p in panels)
        heavy = np.mean([np.mean(p > 3) for p in panels])
        q = per[1:]
        print("   per abscissa (first one excluded): errbd %.2f truncation %.2f doublings %.2f bisections %.2f"
              % tuple(q[:, k].mean() for k in range(4)))
        if os.environ.get("DV_VERBOSE"):
            print("   per-panel max:", [int(p.max()) for p in panels], " position of max:", [int(p.argmax()) for p in panels])
            print("   sorted top work:", np.sort(work)[-8:], " doublings/bisections at the top:", per[np.argmax(work), 2:4])
        print("gene %d: %d abscissae in %d panels; evaluations per panel: mean lane %.1f, max lane %.1f (x%.1f); "
              "abscissae with > 3 evaluations: %.0f %%; aux integrations %d"
              % (g, n, len(panels), mean_sum / len(panels), max_sum / len(panels), max_sum / max(mean_sum, 1e-9),
                 100 * heavy, int(per[:, 4].sum())))


if __name__ == "__main__":
    main()
