#!/usr/bin/env python3
"""Throughput of the MetaCov covariance band (rvt_cov_block) at full size: one block of V variants, N samples.
Reports ms per block, algorithmic bytes/flops (8 N V read once; 2 N V (V/2 + d) flop for the upper triangle) and
covariance pairs per second.  usage (GPU box): python tools/bench_metacov.py [--samples 500000] [--variants 1024]"""
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
    ap.add_argument("--variants", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, V = a.samples, a.variants
    eng = rvtests_amd.Engine(0)
    ld = eng.padded_ld(N)
    X, y, res, sigma2 = bench.fit_null_qt(dev, N, 7)
    eng.set_null(rvtests_amd.TRAIT_QUANTITATIVE, np.asfortranarray(X.cpu().numpy()), res.cpu().numpy().copy(),
                 np.full(N, float(sigma2)), float(sigma2))
    blocks, Ms, afs = bench.make_genes(dev, N, ld, 1, 5, V, V)
    torch.cuda.synchronize()
    # line 1: the fp64 band.  The synthetic block holds hard calls, which the engine would send to the exact int8 product
    # (round 3's line 1 did exactly that under an "fp64" label: 90 "fp64 TFLOP/s", above the 78.6 peak) — so the hard-call
    # path is switched OFF for this measurement (rvt_set_hardcall(0)): gene_suffstat_mfma / gene_suffstat_panel + cov kernels
    eng.set_hardcall(False)
    eng.cov_block(blocks[0].data_ptr(), V)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        cov, xz, zz, poly = eng.cov_block(blocks[0].data_ptr(), V)
    dt = (time.perf_counter() - t0) / a.reps
    eng.set_hardcall(True)
    pairs = V * (V + 1) / 2
    print({"N": N, "V": V, "kernel": "fp64 matrix cores (hard-call path off: what a block of dosages takes)", "ms_per_block": 1e3 * dt,
           "pairs_per_s": pairs / dt, "alg_GBps": 8.0 * N * V / dt / 1e9,
           "alg_TFLOPs_fp64": 2.0 * N * V * (V / 2 + 4) / dt / 1e12, "polymorphic": int(poly.sum())})
    hard = torch.round(blocks[0]).contiguous()
    assert eng.classify_block(hard.data_ptr(), V)
    eng.cov_block(hard.data_ptr(), V)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        cov, xz, zz, poly = eng.cov_block(hard.data_ptr(), V)
    dt = (time.perf_counter() - t0) / a.reps
    print({"N": N, "V": V, "kernel": "hard calls: exact int8 product", "ms_per_block": 1e3 * dt, "pairs_per_s": pairs / dt,
           "alg_GBps": 8.0 * N * V / dt / 1e9, "int8_TOPs": 2.0 * N * V * V / dt / 1e12, "polymorphic": int(poly.sum())})


if __name__ == "__main__":
    main()
