#!/usr/bin/env python3
"""PCIe-inclusive rate of the ModelFitter-style streaming interface: host genotype buffers handed to rvt_submit_gene
(which copies them before returning), results collected every `--window` genes.
usage (GPU box): python tools/bench_stream.py [--samples 500000] [--variants 50] [--genes 128] [--window 64]"""
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
    ap.add_argument("--genes", type=int, default=128)
    ap.add_argument("--window", type=int, default=64)
    ap.add_argument("--packed", action="store_true", help="int8 hard calls through rvt_submit_gene_i8")
    ap.add_argument("--bed", action="store_true", help="PLINK 2-bit codes through rvt_submit_gene_bed")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N = a.samples
    eng = rvtests_amd.Engine(0)
    ld = eng.padded_ld(N)
    X, y, res, sigma2 = bench.fit_null_qt(dev, N, 7)
    eng.set_null(rvtests_amd.TRAIT_QUANTITATIVE, np.asfortranarray(X.cpu().numpy()), res.cpu().numpy().copy(),
                 np.full(N, float(sigma2)), float(sigma2))
    blocks, Ms, afs = bench.make_genes(dev, N, ld, 4, 21, a.variants, a.variants)
    host = [np.asfortranarray(b[:, :N].T.cpu().numpy()) for b in blocks]        # pageable caller buffers
    if a.bed:
        host = [eng.pack_bed(np.rint(h)) for h in host]      # hard calls (imputed means of the synthetic genes rounded)
    elif a.packed:
        host = [np.asfortranarray(h.astype(np.int8)) for h in host]
    del blocks
    for w in (1, a.window):
        t0 = time.perf_counter()
        done = 0
        for g in range(a.genes):
            if a.bed:
                eng.submit_gene_bed(g, host[g % 4], a.variants, want_af=False)
            elif a.packed:
                eng.submit_gene_raw(g, host[g % 4], want_af=False)
            else:
                eng.submit_gene(g, host[g % 4], afs[g % 4])
            if (g + 1) % w == 0:
                done += len(eng.collect())
        done += len(eng.collect())
        dt = time.perf_counter() - t0
        print({"N": N, "M": a.variants, "window": w, "genes": done, "gene_sets_per_s": done / dt,
               "host_GBps": done * (0.25 if a.bed else 1.0 if a.packed else 8.0) * N * a.variants / dt / 1e9,
               "packed": "bed" if a.bed else a.packed})


if __name__ == "__main__":
    main()
