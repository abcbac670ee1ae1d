#!/usr/bin/env python3
"""Throughput of the MetaCov score-covariance band (`--meta cov`; MetaCovTest, src/Model.cpp:844-1004) through the C ABI,
measured like the headline: one JSON line per workload with `roofline` and `cpu_baseline` objects.

  block   one block of V variants, N samples, all pairs of the upper triangle (rvt_cov_block):
            fp64   dosages / anything that is not a hard call: the LDS-tiled fp64 product (gemm_f64.hip.h)  — matrix-core bound
            hc     hard calls under an unweighted model: the exact int8 product (rot_gemm.hip.h)            — HBM bound
  window  the reference's 1 Mb sliding window as the adapter drives it (ModelFitterGpu.cpp MetaCovTest::fit / flush): a stream
          of variants whose window holds `--window` markers; the circular device ring (4 096 columns, doubled until it holds
          four windows) is filled from HBM-resident columns and flushed when full: ONE rvt_cov_band call computes the band of the
          finished heads where the columns lie (modulo the capacity), the head index advances, nothing is moved.

Algorithmic work: per pair of the band 2 N flop; per block 8 N V bytes read once.
CPU baseline: the oracle's MetaCov (orc.metacov, float32 storage as the reference) on a bounded sample (N, V scaled down),
single thread, rate in pairs/s scaled by N (its cost is N per pair).

usage (GPU box): python tools/bench_metacov.py [--samples 500000] [--variants 1024] [--window 200,1000] [--no-cpu]"""
import argparse
import ctypes as C
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
import bench  # noqa: E402

FP64_MATRIX_PEAK_TFLOPS = 78.6   # vendor figure for fp64 matrix (= 32 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz); the guide lists no fp64 row
HBM_PEAK_GBS = 8000.0
FP4_PEAK_TOPS = 10000.0          # MI355X_MICROARCH.md: FP4 MFMA ~10 PF dense (block-scaled mfma_scale_*_f8f6f4)


def cpu_baseline(N_full, use_float=1):
    """orc.metacov on N = 20 000, V = 160 (one window): seconds per pair, scaled to N_full (the cost is linear in N)."""
    import orc
    rng = np.random.default_rng(3)
    N, V, d = 20000, 160, 3
    G = np.asfortranarray(rng.binomial(2, 0.2, size=(N, V)).astype(np.float64))
    X = np.column_stack([np.ones(N), rng.normal(size=(N, 2))])
    y = rng.normal(size=N)
    chrom = np.ones(V, dtype=np.int32)
    pos = np.arange(V, dtype=np.int32)
    t0 = time.perf_counter()
    rc, kept, cov, row_end, xz, zz = orc.metacov(G, chrom, pos, X, y, 0, 10 ** 6, use_float=bool(use_float))
    dt = time.perf_counter() - t0
    pairs = int(np.isfinite(cov).sum())
    per_pair_full = dt / pairs * (N_full / N)
    return {"value": 1.0 / per_pair_full, "unit": "covariance pairs/s at N=%d" % N_full, "cores": 1, "kind": "port",
            "sample": "orc.metacov (float32 storage as the reference), N=%d, V=%d, %d pairs in %.2f s, scaled by N" % (N, V, pairs, dt)}


def stream_window(eng, src_blk, V, N, w, cap, stream):
    """Drive the circular ring as the adapter does: fill from the resident columns of src_blk (device copies that hand on what
    the engine keeps per uploaded column; untimed), one rvt_cov_band call per flush.  The first flush is a warm-up.
    -> dict(done, flushes, calls, t_cov, t_fill, path)"""
    ring = eng.alloc_block(cap)
    heads = cap - w                                           # heads whose window is complete when the ring is full
    if heads > 256:
        heads -= heads % 256                                  # (whole row panels of the band kernel; the rest waits, as in flush())
    band = np.zeros((heads, w + 1), dtype=np.float32)
    eng.host_register(band)                                   # the band lands by DMA (the adapter registers its buffer too)
    scale = np.float32(1.0 / N)
    done = fill = nxt = head = 0
    t_cov = t_fill = 0.0
    flushes = -1
    calls = 0
    while done < stream:
        t1 = time.perf_counter()
        while fill < cap:
            tail = (head + fill) % cap
            n = min(cap - fill, V - nxt, cap - tail)
            eng._check(eng.L.rvt_block_copy_columns(eng.ctx, C.c_void_p(ring), tail, C.c_void_p(src_blk), nxt, n))
            fill += n
            nxt = (nxt + n) % V
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        eng.cov_band(ring, cap, head, heads, cap, w, scale=scale, band=band)
        calls += 1
        t3 = time.perf_counter()
        head = (head + heads) % cap                           # the finished heads are dropped: the head index advances
        fill = cap - heads
        if flushes < 0:                                       # the first flush of a width is a warm-up (first touch of the ring)
            flushes = calls = 0
            continue
        t_fill += t2 - t1
        t_cov += t3 - t2
        flushes += 1
        done += heads
    path = eng.cov_band_last_path()
    eng.host_unregister(band)
    eng.free_block(ring)
    return dict(done=done, flushes=flushes, calls=calls, t_cov=t_cov, t_fill=t_fill, path=path)


def ring_policy(w, forced=0):
    cap = 4096
    while cap < 4 * w:
        cap *= 2
    return max(forced, w + 1) if forced else cap


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=500000)
    ap.add_argument("--variants", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--window", default="200,1000,3000", help="comma-separated window widths (markers) of the stream runs; empty = none")
    ap.add_argument("--stream", type=int, default=40000, help="variants of a stream run")
    ap.add_argument("--ring", type=int, default=0, help="columns of the device ring of the stream runs (default: the adapter's policy)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--skip-blocks", action="store_true", help="only the stream runs (profiles of the window)")
    ap.add_argument("--window-imputed", default="200,1000,3000", help="window widths of the stream runs on MEAN-IMPUTED hard calls")
    ap.add_argument("--missing-rate", type=float, default=0.01)
    ap.add_argument("--window-dosage", default="1000", help="window widths of the DOSAGE stream runs (the fp64 band: what dosages and columns "
                    "with more than one value besides 0 / 1 / 2 take); empty = none")
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
    lines = []   # (printed at the end: the CPU baseline is timed AFTER the device measurements, so that nothing of it —
                 #  threads, heap state — sits under them)
    pairs = V * (V + 1) / 2

    def timed(fn):
        """median of `reps` calls after one warm-up (the GPU boxes are shared hosts: a mean of three carried the occasional
        10 ms stall of a HIP call into a 2 ms figure)"""
        fn()
        ts = []
        for _ in range(max(a.reps, 1)):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], out

    # ---- block, fp64: dosages (the synthetic block + a fractional part in the non-zero entries)
    dos = blocks[0].clone()
    dos += (dos > 0) * 0.125 * torch.rand_like(dos)
    torch.cuda.synchronize()
    eng.set_content_hint(0)                                   # as an adapter reading --dosage input says
    if not a.skip_blocks:
        dt, (cov, xz, zz, poly) = timed(lambda: eng.cov_block(dos.data_ptr(), V))
    eng.set_content_hint(-1)
    del dos
    if not a.skip_blocks:
      tf = 2.0 * N * pairs / dt / 1e12
      lines.append(({"workload": "MetaCov block, dosages (fp64 matrix cores)", "N": N, "V": V, "ms_per_block": 1e3 * dt,
                      "value": pairs / dt, "unit": "covariance pairs/s", "polymorphic": int(poly.sum()),
                      "roofline": {"kernel": "gemm_tn_f64_kernel", "bound": "mfma", "achieved": tf, "peak": FP64_MATRIX_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": tf / FP64_MATRIX_PEAK_TFLOPS, "traffic": None,
                                   "note": "2 N flop per pair of the upper triangle over the wall time of the synchronous C call "
                                           "(column pass, product, reduction, band, copy-back included)"},
                      "cpu_baseline": None}))
    # ---- block, hard calls
    hard = torch.round(blocks[0]).contiguous()
    torch.cuda.synchronize()                                  # (torch's stream is not the engine's)
    assert eng.classify_block(hard.data_ptr(), V)
    if not a.skip_blocks:
      dt, (cov, xz, zz, poly) = timed(lambda: eng.cov_block(hard.data_ptr(), V))
      gbs = 8.0 * N * V / dt / 1e9
      lines.append(({"workload": "MetaCov block, hard calls (exact int8 product)", "N": N, "V": V, "ms_per_block": 1e3 * dt,
                      "value": pairs / dt, "unit": "covariance pairs/s", "polymorphic": int(poly.sum()),
                      "int8_TOPs": 2.0 * N * V * V / dt / 1e12,
                      "roofline": {"kernel": "cov_hc_prep_kernel + rot_gemm_i8_kernel", "bound": "hbm", "achieved": gbs,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                                   "note": "8 N V bytes of the block over the wall time of the synchronous C call"},
                      "cpu_baseline": None}))
    # ---- block, hard calls, filled column by column as the adapter fills its ring: the engine made the int8 copy and the
    #      column statistics behind every upload, the call starts at the integer product
    # V resident columns, re-used cyclically as the stream.  They are put on the device the way the adapter puts a site's
    # column there (rvt_block_upload_columns), so that they carry what the engine keeps per uploaded column — content flag,
    # int8 copy, column statistics — and the device copies below hand that on to the ring (untimed, like the fill)
    src_blk = eng.alloc_block(V)
    hard_host = hard[:, :N].cpu().numpy()                     # (V, N): row j = column j of the block
    for j0 in range(0, V, 64):
        eng.upload_columns(src_blk, j0, np.asfortranarray(hard_host[j0:j0 + 64].T))
    del hard_host
    if not a.skip_blocks:
      dt, (cov2, xz2, zz2, poly2) = timed(lambda: eng.cov_block(src_blk, V))
      assert np.array_equal(np.triu(cov2), np.triu(cov)) and np.array_equal(xz2, xz)       # (same numbers as the block above)
      gbs = 8.0 * N * V / dt / 1e9
      lines.append(({"workload": "MetaCov block, hard calls, block filled by rvt_block_upload_columns (column cache)", "N": N, "V": V,
                   "ms_per_block": 1e3 * dt, "value": pairs / dt, "unit": "covariance pairs/s", "polymorphic": int(poly2.sum()),
                   "roofline": {"kernel": "rot_gemm_i8_kernel", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                                "note": "8 N V bytes of the block over the wall time of the synchronous C call; the call itself reads "
                                        "the N V bytes of the int8 copy made at upload time"},
                   "cpu_baseline": None}))
    # ---- the sliding window, as the adapter drives it (MetaCovTest::fit / flush of ModelFitterGpu.cpp): a CIRCULAR ring — a new
    # site goes into the physical column behind the tail, a flush computes the band of the finished heads where the columns lie
    # (rvt_cov_band addresses them modulo the capacity) and advances the head index; nothing is ever moved.  The ring starts at
    # 4 096 columns and doubles while a flush emits less than three quarters of it (--ring overrides the capacity).
    widths = [int(w) for w in a.window.split(",") if w]
    for w in widths:
        cap = ring_policy(w, a.ring)
        r = stream_window(eng, src_blk, V, N, w, cap, a.stream)
        assert r["path"] in (1, 11), r["path"]                # the integer band on the column cache
        done, flushes, calls, t_cov, t_fill = r["done"], r["flushes"], r["calls"], r["t_cov"], r["t_fill"]
        dt = t_cov
        npairs = done * (w + 1)
        gbs = 8.0 * N * done / dt / 1e9
        lines.append(({"workload": "MetaCov sliding window, hard calls, circular device ring (rvt_cov_band)", "N": N, "window_markers": w,
                          "ring_columns": cap, "variants": done, "flushes": flushes, "device_calls": calls,
                          "ms_per_flush": 1e3 * dt / flushes,
                          "ms_per_flush_in_cov_calls": 1e3 * t_cov / flushes, "ms_per_flush_moving_the_ring": 0.0,
                          "ms_per_flush_filling_the_ring_untimed": 1e3 * t_fill / flushes, "value": npairs / dt,
                          "unit": "printed covariance pairs/s", "variants_per_s": done / dt,
                          "int8_band_TOPs": 2.0 * N * npairs / dt / 1e12,
                          "roofline": {"kernel": "band_gemm_i8_kernel", "bound": "hbm", "achieved": gbs,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                                       "note": "8 N bytes per evicted variant (SURVEY 8d: each column of the stream read once) over "
                                               "the time of the device calls; the calls themselves read the int8 copies (N bytes "
                                               "per column and pass) — int8_band_TOPs = 2 N per printed pair against the 3.94 POP/s "
                                               "int8 ceiling is the figure that bounds this kernel"},
                          "cpu_baseline": None}))
    # ---- the same window on MEAN-IMPUTED hard calls — what the reference's matrices hold when a genotype is missing (the
    # column mean in place of it: src/DataConsolidator.h imputeGenotypeToMean): columns uploaded one at a time as the adapter
    # uploads them, so the engine knows each column's one other value; the band is four exact MXFP4 products
    if a.window_imputed:
        isrc = eng.alloc_block(V)
        ih = hard[:, :N].cpu().numpy().copy()
        rs = np.random.RandomState(11)
        for j in range(V):
            miss = rs.rand(N) < a.missing_rate
            if miss.any() and not miss.all():
                ih[j, miss] = ih[j, ~miss].mean()
        for j in range(V):
            eng.upload_columns(isrc, j, np.asfortranarray(ih[j:j + 1].T))
        del ih
        for w in [int(x) for x in a.window_imputed.split(",") if x]:
            cap = ring_policy(w, a.ring)
            r = stream_window(eng, isrc, V, N, w, cap, a.stream)
            assert r["path"] == 4, r["path"]
            tp = 4 * 2.0 * N * r["done"] * (w + 1) / r["t_cov"] / 1e12
            lines.append({"workload": "MetaCov sliding window, MEAN-IMPUTED hard calls (four MXFP4 products), circular device ring "
                                      "(rvt_cov_band)", "N": N, "missing_rate": a.missing_rate,
                          "window_markers": w, "ring_columns": cap, "variants": r["done"], "flushes": r["flushes"],
                          "ms_per_flush": 1e3 * r["t_cov"] / r["flushes"], "value": r["done"] * (w + 1) / r["t_cov"],
                          "unit": "printed covariance pairs/s", "variants_per_s": r["done"] / r["t_cov"], "fp4_band_TOPs": tp,
                          "roofline": {"kernel": "band_gemm_i8_kernel<FP4>", "bound": "mfma", "achieved": tp, "peak": FP4_PEAK_TOPS,
                                       "unit": "TOP/s", "frac": tp / FP4_PEAK_TOPS, "traffic": None,
                                       "note": "4 x 2 N operations per printed pair (h'h, h'm, m'h, m'm) over the time of the device "
                                               "calls"},
                          "cpu_baseline": None})
        eng.free_block(isrc)
    # ---- the same window on columns that are NOT hard calls (dosages): the
    # band tiles on the fp64 matrix cores (gemm_tn_f64_kernel with the ring's addressing), passes of 1 024 heads
    if a.window_dosage:
        dsrc = eng.alloc_block(V)
        dh = (torch.round(blocks[0]) + (torch.round(blocks[0]) > 0) * 0.125 * torch.rand_like(blocks[0]))[:, :N].cpu().numpy()
        for j0 in range(0, V, 64):
            eng.upload_columns(dsrc, j0, np.asfortranarray(dh[j0:j0 + 64].T))
        del dh
        eng.set_content_hint(0)
        for w in [int(x) for x in a.window_dosage.split(",") if x]:
            cap = 4096
            while cap < 4 * w:
                cap *= 2
            ring = eng.alloc_block(cap)
            heads = cap - w
            heads -= heads % 256
            band = np.zeros((heads, w + 1), dtype=np.float32)
            eng.host_register(band)
            done = fill = nxt = head = 0
            t_cov = 0.0
            flushes = -1
            while done < min(a.stream, 3 * heads):
                while fill < cap:
                    tail = (head + fill) % cap
                    n = min(cap - fill, V - nxt, cap - tail)
                    eng._check(eng.L.rvt_block_copy_columns(eng.ctx, C.c_void_p(ring), tail, C.c_void_p(dsrc), nxt, n))
                    fill += n
                    nxt = (nxt + n) % V
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                eng.cov_band(ring, cap, head, heads, cap, w, scale=np.float32(1.0 / N), band=band)
                t3 = time.perf_counter()
                head = (head + heads) % cap
                fill = cap - heads
                if flushes < 0:
                    flushes = 0
                    continue
                t_cov += t3 - t2
                flushes += 1
                done += heads
            tf = 2.0 * N * done * (w + 1) / t_cov / 1e12
            lines.append({"workload": "MetaCov sliding window, DOSAGES (fp64 band), circular device ring (rvt_cov_band)", "N": N,
                          "window_markers": w, "ring_columns": cap, "variants": done, "flushes": flushes,
                          "ms_per_flush": 1e3 * t_cov / flushes, "value": done * (w + 1) / t_cov, "unit": "printed covariance pairs/s",
                          "variants_per_s": done / t_cov,
                          "roofline": {"kernel": "gemm_tn_f64_kernel", "bound": "mfma", "achieved": tf, "peak": FP64_MATRIX_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": tf / FP64_MATRIX_PEAK_TFLOPS, "traffic": None,
                                       "note": "2 N flop per printed pair over the time of the device calls (column pass, band tiles "
                                               "of the fp64 product, reduction of the K slices, rows, copy-back)"},
                          "cpu_baseline": None})
            eng.host_unregister(band)
            eng.free_block(ring)
        eng.set_content_hint(-1)
        eng.free_block(dsrc)
    cpu = None if a.no_cpu else cpu_baseline(N)
    for ln in lines:
        ln["cpu_baseline"] = cpu
        print(json.dumps(ln))


if __name__ == "__main__":
    main()
