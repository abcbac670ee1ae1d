#!/usr/bin/env python3
"""Time the SKAT permutation test at full size (one gene, default nPerm = 10000, alpha = 0.05).
usage (GPU box): python tools/bench_perm.py [--samples 500000] [--variants 50]"""
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
    ap.add_argument("--variants", type=int, default=50)
    ap.add_argument("--nperm", type=int, default=10000)
    ap.add_argument("--exact", action="store_true", help="replay the reference's rand() stream (rvt_set_perm_exact)")
    ap.add_argument("--genes", type=int, default=2)
    ap.add_argument("--alpha", type=float, default=0.05, help="alpha = 1 disables the adaptive stop: nperm shuffles per gene")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N = a.samples
    eng = rvtests_amd.Engine(0)
    ld = eng.padded_ld(N)
    X, y, res, sigma2 = bench.fit_null_qt(dev, N, 7)
    eng.set_null(rvtests_amd.TRAIT_QUANTITATIVE, np.asfortranarray(X.cpu().numpy()), res.cpu().numpy().copy(),
                 np.full(N, float(sigma2)), float(sigma2))
    blocks, Ms, afs = bench.make_genes(dev, N, ld, a.genes, 11, a.variants, a.variants)
    torch.cuda.synchronize()
    eng.set_perm_exact(a.exact)
    prm = rvtests_amd.Params(1.0, 25.0, 1.0, 25.0, a.nperm, a.alpha)
    t_all = time.perf_counter()
    total = 0
    for k in range(a.genes):
        t0 = time.perf_counter()
        out = eng.run_blocks([blocks[k].data_ptr()], [Ms[k]], [afs[k]], tests=rvtests_amd.TEST_SKAT, params=prm)
        dt = time.perf_counter() - t0
        r = out[0]
        total += r.perm_actual_perm
        if k < 4:
            print({"N": N, "M": Ms[k], "seconds": dt, "actual_perm": r.perm_actual_perm, "num_greater": r.perm_num_greater,
                   "perm_p": r.perm_pvalue, "skat_p": r.skat_p, "perms_per_s": r.perm_actual_perm / dt})
    dt = time.perf_counter() - t_all
    print({"mode": "exact" if a.exact else "counter", "genes": a.genes, "genes_per_s": a.genes / dt,
           "shuffles_per_s": total / dt, "seconds": dt})


if __name__ == "__main__":
    main()
