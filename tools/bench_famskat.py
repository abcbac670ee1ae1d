#!/usr/bin/env python3
"""FamSKAT throughput (SURVEY config 5 shape, reduced N): nuclear families of 4, dense U/S at the boundary.
usage (GPU box): python tools/bench_famskat.py [--samples 20000] [--genes 256] [--variants 30]
Prints the time of the kinship install, the FastLMM null fit and genes/s of rvt_run_fam_blocks, with the
algorithmic work of the rotation GEMM (2 N^2 M flop per gene, U read once per batch)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rvtests_amd  # noqa: E402

HBM_PEAK_GBS = 8000.0
INT8_DENSE_PEAK_TOPS = 3944.0   # MI355X_MICROARCH.md: I8 micro-benchmark ceiling (no spec figure; ~2x the bf16 rate)


def cpu_baseline_literal(M):
    """B-lit-FamSkat: the oracle's LITERAL FamSkat::FitNullModel + TestCovariate — N x N Sigma, Sigma^-1, P0 in
    float as regression/FamSkat.cpp:34-138 forms them — on one gene at N = 400, 800, 1200, one thread; the fitted power law
    says what the reference's formulation costs at the benchmark's N (it cannot run there: 4 N^2 bytes x 3 matrices)."""
    import orc
    rng = np.random.default_rng(4)
    blk = np.array([[1, 0, .5, .5], [0, 1, .5, .5], [.5, .5, 1, .5], [.5, .5, .5, 1]])
    s4, u4 = np.linalg.eigh(blk)
    pts = []
    for N in (400, 800, 1200):
        U = np.zeros((N, N))
        for f in range(N // 4):
            U[4 * f:4 * f + 4, 4 * f:4 * f + 4] = u4
        S = np.tile(s4, N // 4)
        X = np.column_stack([np.ones(N), rng.standard_normal(N)])
        y = rng.standard_normal(N)
        G = np.asfortranarray(rng.binomial(2, 0.05, size=(N, M)).astype(float))
        rc, nul = orc.fastlmm_null(X, y, U, S, use_float=True)
        t0 = time.perf_counter()
        rc2, r = orc.famskat(G, X, y, U, S, nul, use_float=True)
        pts.append((N, time.perf_counter() - t0))
    ex = float(np.polyfit(np.log([p[0] for p in pts]), np.log([p[1] for p in pts]), 1)[0])
    return {"value": 1.0 / pts[-1][1], "unit": "gene-sets/s at N=%d" % pts[-1][0], "cores": 1, "kind": "port",
            "sample": "orc.famskat (literal N x N, float), M=%d, one gene at N = %s: %s s; fitted cost ~ N^%.2f"
                      % (M, [p[0] for p in pts], ["%.2f" % p[1] for p in pts], ex),
            "exponent": ex}


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
    ap.add_argument("--no-cpu", action="store_true")
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
    # algorithmic work: bytes = 4 N^2 (the N x N operand once per batch) + 8 N M per gene; the rotation is
    # 2 N^2 flop per polymorphic column (int8 digit planes: 6 planes of U x 1 plane of hard calls)
    alg_bytes = 4.0 * N * N + 8.0 * N * a.variants * a.genes
    if a.dense:
        tops = 2.0 * N * N * npoly * 6 / dt / 1e12
        roof = {"kernel": "rot_gemm_i8_kernel", "bound": "mfma", "achieved": tops, "peak": INT8_DENSE_PEAK_TOPS, "unit": "TOP/s",
                "frac": tops / INT8_DENSE_PEAK_TOPS, "traffic": None,
                "note": "2 N^2 ops per polymorphic column and digit plane of U (6) over the wall time of the batch"}
    else:
        gbs = alg_bytes / dt / 1e9
        # what the structured rotation actually READS: the share of U its K ranges cover (6 digit planes of one byte) + the genes
        visits = eng.kinship_structure()
        read_bytes = 6.0 * N * N * visits + 8.0 * N * a.variants * a.genes
        roof = {"kernel": "rot_gemm_i8_short_kernel / rot_sparse_kernel + gene_suffstat_mfma", "bound": "hbm", "achieved": gbs,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "frac_is": "NOMINAL", "traffic": None,
                "bytes_actually_read_GBps": read_bytes / dt / 1e9, "frac_of_bytes_actually_read": read_bytes / dt / 1e9 / HBM_PEAK_GBS,
                "note": "frac is NOMINAL: 4 N^2 + 8 N M bytes per gene of the reference's formulation (SURVEY 8d) over the wall time — "
                        "the family-structured rotation visits only the share `rotation_visits` of U, so this is not a bandwidth; "
                        "bytes_actually_read_GBps counts the digit planes of U it does read plus the genes' blocks"}
    print(json.dumps({"workload": "FamSKAT (BASELINE configs[4] shape), %s" % ("dense U" if a.dense else ("families interleaved" if a.shuffle else "family-structured U")),
                      "N": N, "genes": a.genes, "M": a.variants, "rotation_visits": eng.kinship_structure(), "kinship_install_s": t_kin,
                      "null_fit_s": t_null, "delta": nul.delta, "brent_evals": nul.brent_evals, "ms_per_batch": 1e3 * dt,
                      "value": a.genes / dt, "unit": "gene-sets/s", "rotation_TFLOPs_if_all_time": 2.0 * N * N * npoly / dt / 1e12,
                      "ok": sum(r.famskat_ok for r in out), "roofline": roof,
                      "cpu_baseline": None if a.no_cpu else cpu_baseline_literal(a.variants)}))


if __name__ == "__main__":
    main()
