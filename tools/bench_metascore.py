#!/usr/bin/env python3
"""Throughput of the MetaScore single-variant statistics (rvt_score_block) at full size: one device block of V
variants, N samples, read once (8 N V algorithmic bytes).  Reports ms per block, variants per second and the
algorithmic bandwidth.  usage (GPU box): python tools/bench_metascore.py [--samples 500000] [--variants 4096]"""
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
    ap.add_argument("--variants", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
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
    def timed(ptr):                                          # median of single calls: the pool's boxes are shared
        eng.score_block(ptr, V)
        ts = []
        for _ in range(max(a.reps, 7)):
            t0 = time.perf_counter()
            out = eng.score_block(ptr, V)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), out

    dt, r = timed(blocks[0].data_ptr())
    print({"N": N, "V": V, "kernel": "general fp64 (content of the block unknown)", "ms_per_block": 1e3 * dt,
           "variants_per_s": V / dt, "alg_GBps": 8.0 * N * V / dt / 1e9, "tested": int(r["ok"].sum())})
    # the same block as hard calls whose content is known (rvt_block_classify; the adapters get it per column for free
    # when they upload): slices of 32 columns through the int8 kernel
    hard = torch.round(blocks[0]).contiguous()
    assert eng.classify_block(hard.data_ptr(), V)
    dt, r = timed(hard.data_ptr())
    print({"N": N, "V": V, "kernel": "hard-call int8", "ms_per_block": 1e3 * dt, "variants_per_s": V / dt,
           "alg_GBps": 8.0 * N * V / dt / 1e9, "tested": int(r["ok"].sum())})
    # the same hard calls as a RESIDENT .bed matrix (rvt_score_bed_dev): N/4 bytes per site instead of 8 N
    packed = eng.pack_bed(np.asfortranarray(hard[:256, :N].T.cpu().numpy()))
    rows = 32768
    d_bed = eng.bed_alloc(rows)
    for r0 in range(0, rows, packed.shape[0]):
        eng.bed_upload(d_bed, r0, packed[:min(packed.shape[0], rows - r0)])
    eng.score_bed_dev(d_bed, rows, want_counts=False)
    ts = []
    for _ in range(max(a.reps, 7)):
        t0 = time.perf_counter()
        out = eng.score_bed_dev(d_bed, rows)
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))                                 # (the pool's boxes are shared: single calls scatter by 2x)
    print({"N": N, "V": rows, "kernel": "resident .bed rows (gene_tnull_hcp<2, score>)", "ms_per_call": 1e3 * dt,
           "ms_per_call_min_max": [1e3 * min(ts), 1e3 * max(ts)],
           "variants_per_s": rows / dt, "device_GBps_of_codes": (N / 4.0) * rows / dt / 1e9,
           "alg_GBps_at_8N_per_site": 8.0 * N * rows / dt / 1e9, "tested": int(out[0].sum())})
    eng.bed_free(d_bed)


if __name__ == "__main__":
    main()
