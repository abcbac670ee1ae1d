#!/usr/bin/env python3
"""Where the cycles of gene_pvalue_kernel go (needs the profiling build of the engine:
hipcc -DRVT_PROF_K4 ... -> rvtests_amd/csrc/librvtests_amd_prof.so, see tools/build_prof.sh).  One isolated batch.

usage (GPU box): RVT_LIBRARY=rvtests_amd/csrc/librvtests_amd_prof.so python tools/pv_prof.py [--genes 128]
"""
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
    ap.add_argument("--genes", type=int, default=128)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N = a.samples
    eng = rvtests_amd.Engine(0)
    ld = eng.padded_ld(N)
    X, y = bench.make_phenotype(dev, N, 20260002)
    eng.fit_null(0, np.asfortranarray(X.cpu().numpy()), y.cpu().numpy().copy())
    blocks, Ms, afs = bench.make_genes(dev, N, ld, a.genes, 20260002, 20, 80)
    torch.cuda.synchronize()
    for b, M in zip(blocks, Ms):
        eng.classify_block(b.data_ptr(), M)
    bt = eng.prepare([t.data_ptr() for t in blocks], Ms, afs)
    eng.launch(bt)
    eng.sync()
    eng.set_profiling(True)
    eng.timing(reset=True)
    t0 = time.perf_counter()
    eng.launch(bt)
    eng.sync()
    wall = time.perf_counter() - t0
    tm = eng.timing(reset=True)
    out = bt["out"]
    f = lambda name: np.array([getattr(r, name) for r in out])
    tot, dav, post, neval = f("zeg_U"), f("cmc_U"), f("cmc_V"), f("zeg_V")
    front, main, book, pre = f("famcmc_U"), f("famcmc_V"), f("famzeg_U"), f("famzeg_V")
    pre_a, pre_b, pre_c, pre_d = f("famskat_Q"), f("famskat_p"), f("famcmc_af"), f("famzeg_af")
    print("batch of %d genes: wall %.2f ms; pvalue kernel %.2f ms" % (a.genes, wall * 1e3, tm.ms_pvalue))
    for name, v in (("total QAGS loop", tot), ("  davies rounds", dav), ("    front", front), ("    main", main),
                    ("  post (Liu, density)", post), ("  lane-0 bookkeeping", book), ("before the loop", pre),
                    ("  loads, sorted copies", pre_a), ("  per-rho tails", pre_b), ("  quantiles, Liu moments", pre_c),
                    ("  the two preludes", pre_d)):
        print("%-24s mean %10.0f  max %10.0f  cycles   (%.2f / %.2f ms at 2.1 GHz)" % (name, v.mean(), v.max(),
                                                                                     v.mean() / 2.1e6, v.max() / 2.1e6))
    asm = [f(n) for n in ("vt_minmaf", "vt_maxmaf", "vt_optmaf", "vt_U", "vt_V", "vt_stat", "vt_p", "vt_p_error")]
    print("gene_assemble (thread 0's clock at the phase boundaries):")
    for name, v in zip(("  1 partial statistics + column statistics", "  masked-entry corrections + symmetric completion",
                        "  2 polymorphic columns + burden statistics", "  3 flip algebra", "  4 projected matrix + 5 weights",
                        "  6 SKAT Q + 7 SKAT-O row sums", "  7 SKAT-O scalars", "  whole stage"), asm):
        print("%-50s mean %10.0f  max %10.0f  cycles   (%.3f / %.3f ms at 2.1 GHz)" % (name, v.mean(), v.max(), v.mean() / 2.1e6, v.max() / 2.1e6))
    v = f("perm_pvalue")
    print("    of phase 1, thread 0 in the partial-statistics loop: mean %.0f max %.0f cycles; wave-parts %.0f" % (v.mean(), v.max(), f("cmc_stat").mean()))
    print("neval mean %.0f max %.0f; davies terms mean %.0f" % (neval.mean(), neval.max(), f("davies_terms").mean()))
    k = int(np.argmax(tot))
    print("slowest gene: M=%d neval=%d total %.0f front %.0f main %.0f book %.0f" % (Ms[k], neval[k], tot[k], front[k], main[k], book[k]))


if __name__ == "__main__":
    main()
