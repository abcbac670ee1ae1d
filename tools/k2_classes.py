#!/usr/bin/env python3
"""Isolated bandwidth of the sufficient-statistics kernel per tile class (one class per batch, nothing overlapping).

usage (GPU box): python tools/k2_classes.py [--samples 500000] [--genes 64] [--reps 5]
Prints, per class, ms per launch and algorithmic TB/s (8 N (M + d + 2) bytes per gene).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rvtests_amd  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=500000)
    ap.add_argument("--genes", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--ms", type=str, default="28,32,44,48,60,64,76,80")
    ap.add_argument("--tests", type=int, default=rvtests_amd.TEST_SKAT)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N = a.samples
    eng = rvtests_amd.Engine(0)
    ld = eng.padded_ld(N)
    X, y, res, sigma2 = bench.fit_null_qt(dev, N, 7)
    eng.set_null(rvtests_amd.TRAIT_QUANTITATIVE, np.asfortranarray(X.cpu().numpy()), res.cpu().numpy().copy(),
                 np.full(N, float(sigma2)), float(sigma2))
    print("%4s %7s %10s %10s" % ("M", "class", "ms/launch", "TB/s"))
    for M in [int(x) for x in a.ms.split(",")]:
        blocks, Ms, afs = bench.make_genes(dev, N, ld, a.genes, 100 + M, M, M)
        torch.cuda.synchronize()
        b = eng.prepare([t.data_ptr() for t in blocks], Ms, afs, tests=a.tests)
        eng.launch(b)
        eng.sync()
        eng.set_profiling(True)
        eng.timing(reset=True)
        for _ in range(a.reps):
            eng.launch(b)
            eng.sync()
        tm = eng.timing(reset=True)
        eng.set_profiling(False)
        ms = tm.ms_suffstat / max(tm.n_suffstat_launches, 1)
        tbs = tm.alg_bytes / max(tm.n_suffstat_launches, 1) / (ms * 1e-3) / 1e12
        d = 3
        print("%4d (%d,%d) %10.3f %10.3f" % (M, (M + 15) // 16, (M + d + 1 + 15) // 16, ms, tbs))
        del blocks, b
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
