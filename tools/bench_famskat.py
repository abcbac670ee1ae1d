#!/usr/bin/env python3
"""FamSKAT throughput (SURVEY config 5 shape, reduced N): nuclear families of 4, dense U/S at the boundary.
usage (GPU box): python tools/bench_famskat.py [--samples 20000] [--genes 256] [--variants 30]
Prints the time of the kinship install, the FastLMM null fit and genes/s of rvt_run_fam_blocks, with the
algorithmic work of the rotation GEMM (2 N^2 M flop per gene, U read once per batch)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rvtests_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=20000)
    ap.add_argument("--genes", type=int, default=256)
    ap.add_argument("--variants", type=int, default=30)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--shuffle", action="store_true",
                    help="families interleaved in the sample order (random membership): the sparse-gather rotation")
    ap.add_argument("--dense", action="store_true",
                    help="treat U as a dense matrix (RVT_KINSHIP_DENSE=1): the rate of a GRM's eigenvectors")
    a = ap.parse_args()
    N = a.samples // 4 * 4
    rng = np.random.default_rng(4)
    blk = np.array([[1, 0, .5, .5], [0, 1, .5, .5], [.5, .5, 1, .5], [.5, .5, .5, 1]])
    s4, u4 = np.linalg.eigh(blk)
    U = np.zeros((N, N), dtype=np.float32, order="F")
    member = rng.permutation(N) if a.shuffle else np.arange(N)
    for f in range(N // 4):
        U[member[4 * f:4 * f + 4], 4 * f:4 * f + 4] = u4
    S = np.tile(s4, N // 4).astype(np.float32)
    # a random orthogonal mixing inside the eigenspaces is not needed: any orthogonal U with these S is a valid input
    X = np.column_stack([np.ones(N), rng.standard_normal(N), rng.standard_normal(N)])
    fam = np.zeros(N)
    fam[member] = np.repeat(rng.standard_normal(N // 4), 4)
    y = 0.3 * X[:, 1] - 0.2 * X[:, 2] + np.sqrt(0.4) * fam + np.sqrt(0.6) * rng.standard_normal(N)
    if a.dense:
        os.environ["RVT_KINSHIP_DENSE"] = "1"
    eng = rvtests_amd.Engine(0)
    t0 = time.perf_counter()
    eng.set_kinship(U, S)
    t_kin = time.perf_counter() - t0
    t0 = time.perf_counter()
    nul = eng.fit_fam_null(X, y)
    t_null = time.perf_counter() - t0
    dev = torch.device("cuda:0")
    ld = eng.padded_ld(N)
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    blocks = []
    for k in range(a.genes):
        maf = torch.tensor(10 ** rng.uniform(np.log10(5e-4), np.log10(5e-2), a.variants), device=dev,
                           dtype=torch.float32)
        G = torch.zeros((a.variants, ld), dtype=torch.float64, device=dev)
        for h in range(2):
            G[:, :N] += (torch.rand((a.variants, N), generator=g, device=dev) < maf[:, None]).to(torch.float64)
        blocks.append(G)
    torch.cuda.synchronize()
    ptrs = [b.data_ptr() for b in blocks]
    Ms = [a.variants] * a.genes
    out = eng.run_fam_blocks(ptrs, Ms)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = eng.run_fam_blocks(ptrs, Ms)
    dt = (time.perf_counter() - t0) / a.reps
    npoly = sum(r.n_poly for r in out)
    print({"N": N, "genes": a.genes, "M": a.variants, "rotation_visits": eng.kinship_structure(), "kinship_install_s": t_kin, "null_fit_s": t_null,
           "delta": nul.delta, "brent_evals": nul.brent_evals, "ms_per_batch": 1e3 * dt,
           "gene_sets_per_s": a.genes / dt, "rotation_TFLOPs_if_all_time": 2.0 * N * N * npoly / dt / 1e12,
           "ok": sum(r.famskat_ok for r in out)})


if __name__ == "__main__":
    main()
