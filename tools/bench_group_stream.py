#!/usr/bin/env python3
"""From-host rate of the device-group interface (rvt_group_*: several engine contexts behind ONE caller thread) next to a
single context, and the link's own pinned host-to-device rate for reference.  On a 1-GPU box both members sit on device 0
and share its link and its compute units: measured (round 3, 2 048 genes) two members reach 94 % of one member's rate for
fp64 blocks and 67-78 % for the packed hand-offs — what two contexts cost on ONE device, not what a second device adds
(no multi-GPU box was available to measure that).
usage (GPU box): python tools/bench_group_stream.py [--samples 500000] [--variants 50] [--genes 256]"""
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
    ap.add_argument("--genes", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, M = a.samples, a.variants
    # the link itself: pinned host memory to the device, 256 MB pieces
    src = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
    dst = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    print({"pinned_h2d_GBps": 8 * src.numel() / (time.perf_counter() - t0) / 1e9})
    del src, dst
    X, y, res, sigma2 = bench.fit_null_qt(dev, N, 7)
    Xh, yh = np.asfortranarray(X.cpu().numpy()), y.cpu().numpy().copy()
    eng0 = rvtests_amd.Engine(0)
    ld = eng0.padded_ld(N)
    blocks, Ms, afs = bench.make_genes(dev, N, ld, 4, 21, M, M, missing_frac=0.0)
    host = [np.asfortranarray(b[:, :N].T.cpu().numpy()) for b in blocks]
    del blocks
    torch.cuda.empty_cache()
    data = {"fp64": host, "int8": [np.asfortranarray(h.astype(np.int8)) for h in host],
            "bed2bit": [eng0.pack_bed(np.rint(h)) for h in host]}
    per_gene = {"fp64": 8.0 * N * M, "int8": 1.0 * N * M, "bed2bit": 0.25 * N * M}
    eng0.close()
    for members in (1, 2):
        grp = rvtests_amd.Group([0] * members)
        grp.fit_null(0, Xh, yh)
        for mode in ("fp64", "int8", "bed2bit"):
            n = a.genes if mode != "fp64" else max(64, a.genes // 2)
            t0 = None
            done = 0
            for g in range(-64, n):
                if g == 0:
                    grp.collect()
                    t0 = time.perf_counter()
                    done = 0
                d = data[mode][g % 4]
                if mode == "fp64":
                    grp.submit_gene(g, d, afs[g % 4])
                elif mode == "int8":
                    grp.submit_gene_i8(g, d)
                else:
                    grp.submit_gene_bed(g, d, M)
                if (g + 1) % 64 == 0 and g >= 0:
                    done += len(grp.collect_ready())
            done += len(grp.collect())
            dt = time.perf_counter() - t0
            print({"members": members, "mode": mode, "genes": done, "gene_sets_per_s": done / dt,
                   "host_GBps": done * per_gene[mode] / dt / 1e9})
        grp.close()


if __name__ == "__main__":
    main()
