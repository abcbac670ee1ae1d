#!/usr/bin/env python3
"""Throughput of the two widened gene-level tests at full size (N = 500 000): AnalyticVT as one more test bit of the
resident-block batch (RVT_TEST_ANALYTICVT beside SKAT + SKAT-O + CMC + Zeggini), and KBAC (binary trait, no covariates,
default nPerm = 10000 / alpha = 0.05) on a null gene.  usage (GPU box): python tools/bench_vt_kbac.py [--samples 500000]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rvtests_amd  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=500000)
    ap.add_argument("--genes", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N = a.samples
    eng = rvtests_amd.Engine(0)
    ld = eng.padded_ld(N)
    X, y, res, sigma2 = bench.fit_null_qt(dev, N, 7)
    eng.set_null(rvtests_amd.TRAIT_QUANTITATIVE, np.asfortranarray(X.cpu().numpy()), res.cpu().numpy().copy(),
                 np.full(N, float(sigma2)), float(sigma2))
    blocks, Ms, afs = bench.make_genes(dev, N, ld, a.genes, 11, 20, 80)
    for b, M in zip(blocks, Ms):
        eng.classify_block(b.data_ptr(), M)
    eng.reserve(Ms)
    ptrs = [b.data_ptr() for b in blocks]
    for tests, name in ((15, "SKAT+SKATO+CMC+Zeggini"), (15 | 128, "the same + AnalyticVT"), (128, "AnalyticVT alone")):
        eng.run_blocks(ptrs, Ms, afs, tests=tests)
        t0 = time.perf_counter()
        for _ in range(3):
            out = eng.run_blocks(ptrs, Ms, afs, tests=tests)
        dt = (time.perf_counter() - t0) / 3
        extra = {}
        if tests & 128:
            ok = [r for r in out if r.vt_ok]
            extra = {"vt_ok": len(ok), "mean_cutoffs": float(np.mean([r.vt_ncutoff for r in ok])),
                     "max_p_error": float(max(r.vt_p_error for r in ok))}
        print({"N": N, "genes": a.genes, "tests": name, "ms_per_batch": 1e3 * dt, "gene_sets_per_s": a.genes / dt, **extra})
    # KBAC: binary phenotype, intercept only
    yb = (torch.rand(N, device=dev) < 0.4).double().cpu().numpy()
    eng.fit_null(rvtests_amd.TRAIT_BINARY, np.ones((N, 1)), yb)
    k = int(np.argmin(np.abs(np.array(Ms) - 50)))
    eng.rand_seed(1)
    t0 = time.perf_counter()
    r = eng.kbac_blocks([ptrs[k]], [Ms[k]], [afs[k]], yb, 10000, 0.05)[0]
    dt = time.perf_counter() - t0
    print({"N": N, "test": "KBAC nPerm=10000 alpha=0.05", "M": Ms[k], "seconds": dt, "permutations": r.actual_perm,
           "perms_per_s": r.actual_perm / dt, "patterns": r.n_pattern, "carriers": r.n_carrier, "pvalue": r.pvalue})


if __name__ == "__main__":
    main()
