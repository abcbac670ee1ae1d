"""Times rvt_kinship_decompose (KinshipHolder::decompose on the device) on a nuclear-family kinship and on a dense GRM.
usage: python tools/bench_decompose.py [--samples 8000] [--kind family|grm]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rvtests_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=8000)
ap.add_argument("--kind", default="family")
ap.add_argument("--install", action="store_true",
                help="install the decomposition for the family models instead of returning the eigenvectors (large N)")
args = ap.parse_args()
N = args.samples // 4 * 4
rng = np.random.default_rng(1)
if args.kind == "family":
    blk = np.array([[1, 0, .5, .5], [0, 1, .5, .5], [.5, .5, 1, .5], [.5, .5, .5, 1]], dtype=np.float32)
    K = np.zeros((N, N), dtype=np.float32, order="F")
    for f in range(N // 4):
        K[4 * f:4 * f + 4, 4 * f:4 * f + 4] = blk
elif args.kind == "lowrank":
    # a dense matrix that is cheap to make at N = 100 000 (the GRM below needs an N x 2N factor): an equally spaced diagonal
    # plus a dense rank-64 term of the same size — eigenvalues interlace the diagonal's, no clusters; filled a block of columns at a time
    Z = rng.standard_normal((N, 64)).astype(np.float32)
    K = np.zeros((N, N), dtype=np.float32, order="F")
    for c0 in range(0, N, 4096):
        c1 = min(N, c0 + 4096)
        K[:, c0:c1] = (Z @ Z[c0:c1].T) / np.float32(N)
    K[np.arange(N), np.arange(N)] += np.linspace(0.5, 1.5, N, dtype=np.float32)
    for c0 in range(0, N, 4096):                     # exact symmetry (the product is symmetric only to rounding)
        c1 = min(N, c0 + 4096)
        blk = K[c0:c1, :c1].copy()
        K[:c1, c0:c1] = blk.T
else:
    Z = rng.standard_normal((N, 2 * N)).astype(np.float32)
    K = np.asfortranarray((Z @ Z.T) / np.float32(2 * N))
    K = (K + K.T) / 2
eng = rvtests_amd.Engine(0)
if args.install:
    t0 = time.perf_counter()
    eng.kinship_decompose(K, install=True, want_vectors=False)
    dt = time.perf_counter() - t0
    print(json.dumps({"N": N, "kind": args.kind, "seconds_decompose_and_install": dt,
                      "rotation_visits": eng.kinship_structure()}), flush=True)
    eng.close()
    sys.exit(0)
t0 = time.perf_counter()
U, S, info = eng.kinship_decompose(K, want_vectors=True)
dt = time.perf_counter() - t0
# residual on a sample of columns
idx = rng.choice(N, 16, replace=False)
Ud = U[:, idx].astype(np.float64)
res = np.abs(K.astype(np.float64) @ Ud - Ud * S[idx].astype(np.float64)).max()
# Jacobi: 12 n^3 per sweep; through the tridiagonal form: 4/3 n^3 (reduction) + 2 n^3 (back-transformation) + 4 n^3 (check)
flops = 12.0 * info.padded_order ** 3 * info.sweeps if info.sweeps else (4.0 / 3 + 2 + 4) * float(N) ** 3
w = np.linalg.eigvalsh(K.astype(np.float64)) if N <= 16000 else None
# CPU baseline: what KinshipHolder::decompose does on the host — a dense symmetric eigensolver on the FLOAT matrix (Eigen's
# SelfAdjointEigenSolver<MatrixXf> there, LAPACK's ssyevd through numpy here), bounded to n = 4000 and scaled by N^3
nb = min(N, 4000)
t1 = time.perf_counter()
np.linalg.eigh(np.ascontiguousarray(K[:nb, :nb]))
tb = time.perf_counter() - t1
try:
    from threadpoolctl import threadpool_info
    thr = max([p_.get("num_threads", 1) for p_ in threadpool_info()] or [1])
except Exception:
    thr = None
cpu = {"value": tb * (N / nb) ** 3, "unit": "s per decomposition at N=%d" % N, "cores": thr, "kind": "port",
       "sample": "numpy.linalg.eigh (ssyevd) of the leading %d x %d float block: %.2f s, scaled by (N / %d)^3" % (nb, nb, tb, nb)}
print(json.dumps({"N": N, "kind": args.kind, "seconds": dt, "sweeps": info.sweeps,
                  "solver": "jacobi" if info.sweeps else "tridiagonal form",
                  "eigenvalues_vs_lapack_max_abs": None if w is None else float(np.abs(w - S).max()),
                  "orthogonality_sample_max_abs": float(np.abs(Ud.T @ Ud - np.eye(16)).max()), "max_cosine": info.max_cosine,
                  "shift": info.shift, "residual_max_abs": res, "lambda_min": float(S[0]), "lambda_max": float(S[-1]),
                  "fp64_TFLOPs_nominal": flops / dt / 1e12, "cpu_baseline": cpu}))
eng.close()
