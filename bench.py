#!/usr/bin/env python3
"""bench.py — gene-sets/sec of the kernel/burden hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (SKAT + SKAT-O + CMC + Zeggini, analytic p-values) over one batch of
`--genes` synthetic genes that are ALREADY RESIDENT in HBM in the engine's boundary layout (fp64, column-major,
leading dimension padded to 16 samples).  Workload = BASELINE.json configs[2] ("N=500k, M̄=50,
--kernel skat,skato --burden cmc,zeggini"), the configuration the metric is quoted on: per GPU a shard of
genes with M_g ~ Uniform{20..80}; genes shard across ranks with no data-path collective (weak scaling) — the
only collectives are one broadcast of the null model and one gather of the per-gene result records per step.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the engine's stream around the
sufficient-statistics kernel; `cpu_baseline` times the CPU oracle (literal SKAT-O + folded SKAT + CMC + Zeggini,
reference release flags -O2 -msse2) on a sample of genes of the same workload — 1 thread alone, then one process
per core — and `parity` compares the oracle's numbers for those genes with the GPU records of the timed run
(the second half of the BASELINE metric: p-value max-abs-diff).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (torch.distributed.run as a child
process, before this process touches the GPU).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

_CPU_WORKER = len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker"   # child of cpu_baseline_all_cores: CPU only
if not _CPU_WORKER:
    import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import rvtests_amd  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (≈6.3 TB/s achievable)


def make_genes(dev, N, ld, n_genes, seed, m_lo, m_hi, missing_frac=0.05, dosage=False, dosage_float=False):
    """Synthetic genotype blocks on the device (config 3 of SURVEY.md §8d): per-variant MAF ~ LogUniform(5e-4,
    5e-2), g ~ Binomial(2, maf); 0.1 % of genotypes missing in 5 % of the genes and imputed to the column mean
    exactly as DataConsolidator::imputeGenotypeToMean leaves them.  Returns blocks (M x ld tensors, i.e.
    column-major N x M with leading dimension ld), Ms and counter allele frequencies."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    blocks, Ms, afs = [], [], []
    for k in range(n_genes):
        M = int(rng.integers(m_lo, m_hi + 1))
        maf = torch.tensor(10 ** rng.uniform(np.log10(5e-4), np.log10(5e-2), M), device=dev, dtype=torch.float32)
        G = torch.zeros((M, ld), dtype=torch.float64, device=dev)
        for h in range(2):
            u = torch.rand((M, N), generator=g, device=dev, dtype=torch.float32)
            G[:, :N] += (u < maf[:, None]).to(torch.float64)
            del u
        nsample = float(N)
        if dosage:
            # imputed dosages: every call blurred by its genotype probabilities and printed with three decimals, as the
            # DS field of an imputation server's VCF holds it (`rvtest --dosage DS`): the double nearest to K / 1000.  No
            # block of the batch holds hard calls only; imputed data has no missing entries.
            e = torch.rand((M, N), generator=g, device=dev, dtype=torch.float32) * 0.12
            Gv = G[:, :N]
            # (the division by a device TENSOR: torch turns `x / 1000.0` with a Python scalar into x * (1 / 1000.0) on the
            #  GPU, which is up to 1.5 ulp away from the double strtod makes of "0.998" — the lattice kernel's test, exact
            #  to the last bit since round 4, rightly refuses such values)
            den = torch.full((), 1000.0, device=dev, dtype=torch.float64)
            if dosage_float:
                # float-precision dosages as an 8-bit BGEN file gives them: p1 = float(v1) * float(1 / 255), p2 likewise,
                # dosage = p1 + 2 p2 in double (src/BGenGenotypeExtractor.cpp:413-478) — integer multiples of 2^-31
                blur = (Gv * (1.0 - e.to(torch.float64)) + 0.5 * e.to(torch.float64) * (2.0 - Gv))
                v2 = torch.round(torch.clamp(blur - 1.0, min=0.0) * 255.0)
                v1 = torch.round(torch.clamp(blur - 2.0 * v2 / 255.0, min=0.0, max=1.0) * 255.0)
                sc = torch.full((), 1.0 / 255.0, device=dev, dtype=torch.float32)
                p1 = (v1.to(torch.float32) * sc).to(torch.float64)
                p2 = (v2.to(torch.float32) * sc).to(torch.float64)
                Gv.copy_(p1 + 2.0 * p2)
                del blur, v1, v2, p1, p2
            else:
                Gv.copy_(torch.round((Gv * (1.0 - e.to(torch.float64)) + 0.5 * e.to(torch.float64) * (2.0 - Gv)) * 1000.0) / den)
            del e
        if not dosage and rng.random() < missing_frac:
            miss = torch.rand((M, N), generator=g, device=dev, dtype=torch.float32) < 1e-3
            Gv = G[:, :N]
            ac = torch.where(miss, torch.zeros_like(Gv), Gv).sum(1)                  # integer-valued
            an = 2.0 * (~miss).sum(1).to(torch.float64)
            mean = 2.0 * torch.floor(ac) / an
            Gv[miss] = mean[:, None].expand(-1, N)[miss]
            af = 0.5 * ac / nsample                                                   # GenotypeCounter::getAF
            del miss
        else:
            af = 0.5 * G[:, :N].sum(1) / nsample
        blocks.append(G)
        Ms.append(M)
        afs.append(af.cpu().numpy())
    return blocks, Ms, afs


N_CAUSAL = 12          # genes 0 .. 11 of rank 0's shard carry an effect ("10 causal genes")


def causal_effect(blocks, N, binary):
    """Genetic effect of the causal genes: beta_k x (rare alleles in the first 5 variants of gene k), the betas graded so
    that the association p-values of the causal genes spread over the decades 1e-2 .. 1e-14 — the in-run parity check
    then sees small p-values, not only null genes.  A device vector of N doubles."""
    eff = torch.zeros(N, dtype=torch.float64, device=blocks[0].device)
    for k in range(min(N_CAUSAL, len(blocks))):
        burden = blocks[k][:5, :N].sum(0)
        var = float(burden.var())
        ncp = 8.0 + 7.0 * k                                   # non-centrality aimed at: 8 .. 85
        beta = (ncp / max(var * N, 1e-30)) ** 0.5
        eff += (4.0 if binary else 1.0) * beta * (burden - burden.mean())
    return eff


def make_phenotype(dev, N, seed, binary=False, effect=None):
    """Covariates and phenotype of SURVEY §8d: quantitative y = 0.3 c1 - 0.2 c2 + genes + N(0,1) (configs 2/3) or binary
    y ~ Bernoulli(logit^-1(-2 + 0.3 c1 + genes)) (config 4).  Returns X (with intercept) and y as device tensors."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    c = torch.randn((N, 2), generator=g, device=dev, dtype=torch.float64)
    X = torch.cat([torch.ones((N, 1), device=dev, dtype=torch.float64), c], 1)
    e = effect if effect is not None else torch.zeros(N, dtype=torch.float64, device=dev)
    if binary:
        pr = torch.sigmoid(-2.0 + 0.3 * c[:, 0] + e)
        y = (torch.rand(N, generator=g, device=dev, dtype=torch.float64) < pr).to(torch.float64)
    else:
        y = 0.3 * c[:, 0] - 0.2 * c[:, 1] + e + torch.randn(N, generator=g, device=dev, dtype=torch.float64)
    return X, y


def fit_null_qt(dev, N, seed):
    """Quantitative null model y ~ 1 + c1 + c2 solved with torch (used by the tools/ microbenchmarks)."""
    X, y = make_phenotype(dev, N, seed)
    beta = torch.linalg.solve(X.T @ X, X.T @ y)
    res = y - X @ beta
    sigma2 = float((res @ res) / N)
    return X, y, res, sigma2


def oracle_gene(orc, G_host, af, X, y, res, v, binary):
    """The four ModelFitter::fit bodies of the workload on one gene through the CPU oracle: SKAT (P0 folded; the
    literal N x N form cannot run at this N), literal SKAT-O, CMC and Zeggini.  Returns (seconds, numbers)."""
    t0 = time.perf_counter()
    rc1, a = orc.skat(G_host, af, X, res, v, binary)
    rc2, o = orc.skato(G_host, af, X, res, v, binary)
    rc3, c = orc.burden(G_host, X, y, binary, 0)
    rc4, z = orc.burden(G_host, X, y, binary, 1)
    t = time.perf_counter() - t0
    return t, dict(skat_Q=a.Q, skat_p=a.pvalue, skato_Q=o.Q, skato_p=o.pvalue, skato_rho=o.rho, cmc_p=c.pvalue,
                   zeg_p=z.pvalue, cmc_nonref=c.nonref_site, n_poly=a.n_poly,
                   ok=[int(a.fit_ok), int(o.fit_ok), int(c.fit_ok), int(z.fit_ok)])


def cpu_worker(path):
    """Child process of cpu_oracle_pool: CPU only (torch is never imported).  Fits the null model once, then runs its
    genes; prints one JSON object."""
    import orc
    z = np.load(path)
    X = np.asfortranarray(z["X"])
    y = z["y"]
    binary = int(z["binary"])
    if binary:
        rc, beta, p, v = orc.fit_logistic(X, y)
        res = y - p
    else:
        rc, beta, pred, res, s2 = orc.fit_linear(X, y)
        v = np.full(len(y), s2)
    out = []
    t_all = time.perf_counter()
    for k in z["genes"]:
        t, r = oracle_gene(orc, np.asfortranarray(z["G%d" % k]), z["af%d" % k], X, y, res, v, binary)
        r["gene"] = int(k)
        r["seconds"] = t
        out.append(r)
    print(json.dumps({"results": out, "seconds": time.perf_counter() - t_all}))


def literal_skat_baseline(N_full, M=50):
    """orc.skat_literal(use_float=1) — the LITERAL Skat::Fit with its N x N float P0 — on one synthetic gene at three N."""
    import orc
    import synth
    pts = []
    for N in (4000, 8000, 16000):
        rng = np.random.default_rng(N)
        G = np.asfortranarray(rng.binomial(2, 0.02, size=(N, M)).astype(np.float64))
        X, y, res, v, s2 = synth.make_null(N, 3, 0, seed=3)
        t0 = time.perf_counter()
        rc, lit = orc.skat_literal(G, G.sum(0) / (2.0 * N), X, res, v, 0, use_float=1)
        pts.append((N, time.perf_counter() - t0))
    ex = float(np.polyfit(np.log([p[0] for p in pts]), np.log([p[1] for p in pts]), 1)[0])
    at_full = pts[-1][1] * (N_full / pts[-1][0]) ** ex
    return {"value": 1.0 / pts[-1][1], "unit": "gene-sets/s at N=%d (SKAT only)" % pts[-1][0], "cores": 1, "kind": "port",
            "sample": "orc.skat_literal (N x N P0 in float, Skat.cpp:57-76), M=%d, one gene at N = %s: %s s; cost ~ N^%.2f; "
                      "extrapolated to N=%d: %.0f s per gene (and 4 N^2 bytes = %.1f TB)"
                      % (M, [p[0] for p in pts], ["%.2f" % p[1] for p in pts], ex, N_full, at_full, 4.0 * N_full ** 2 / 1e12),
            "exponent": ex, "extrapolated_seconds_per_gene": at_full}


def cpu_oracle_pool(genes, X, y, binary, workers, library=None, replicate=False):
    """Run the oracle on `genes` ({index: (G_host, af)}) in `workers` independent single-thread processes at once, the
    genes dealt round-robin (genes are independent: this is how the CPU port would use a whole host).  The timed
    wall clock starts when the processes are started and includes their start-up and null fit.  Returns
    (records by gene index, wall seconds)."""
    import subprocess
    import tempfile
    idx = sorted(genes)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as td:
        paths = []
        for w in range(workers):
            mine = idx[w::workers]
            if replicate:                       # every process computes the same genes, out of ONE file
                if w > 0:
                    paths.append(paths[0])
                    continue
                mine = idx
            if not mine:
                continue
            path = os.path.join(td, "w%d.npz" % w)
            arrs = dict(X=X, y=y, binary=np.int64(binary), genes=np.array(mine, dtype=np.int64))
            for k in mine:
                arrs["G%d" % k] = genes[k][0]
                arrs["af%d" % k] = genes[k][1]
            np.savez(path, **arrs)
            paths.append(path)
        env = dict(os.environ)
        if library:
            env["ORC_LIBRARY"] = library
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for path in paths]
        recs = {}
        ok = True
        for pr in procs:
            out, _ = pr.communicate()
            if pr.returncode != 0 or not out.strip():
                ok = False
                continue
            for r in json.loads(out.decode().strip().splitlines()[-1])["results"]:
                recs[r["gene"]] = r
        wall = time.perf_counter() - t0
    return (recs if ok else None), wall


def parity_summary(recs, gpu):
    """GPU records vs oracle records of the same genes: the BASELINE metric's parity half."""
    worst = dict(pvalue_max_abs_diff=0.0, pvalue_max_rel_diff=0.0, q_max_rel_diff=0.0)
    nonref_equal = True
    for k, r in recs.items():
        g = gpu[k]
        for name in ("skat_p", "skato_p", "cmc_p", "zeg_p"):
            a, b = getattr(g, name), r[name]
            worst["pvalue_max_abs_diff"] = max(worst["pvalue_max_abs_diff"], abs(a - b))
            worst["pvalue_max_rel_diff"] = max(worst["pvalue_max_rel_diff"], abs(a - b) / max(abs(b), 1e-300))
        for name in ("skat_Q", "skato_Q"):
            a, b = getattr(g, name), r[name]
            worst["q_max_rel_diff"] = max(worst["q_max_rel_diff"], abs(a - b) / max(abs(b), 1e-300))
        nonref_equal = nonref_equal and g.cmc_nonref == r["cmc_nonref"] and g.n_poly == r["n_poly"] and \
            abs(g.skato_rho - r["skato_rho"]) < 1e-12
    worst["genes_compared"] = len(recs)
    worst["counts_bit_exact"] = bool(nonref_equal)
    worst["min_p_compared"] = min(min(r["skat_p"], r["skato_p"]) for r in recs.values()) if recs else None
    # per decade of the oracle's p-value: largest relative difference, SKAT and SKAT-O apart (north_star: 1e-6 relative)
    dec = {}
    for k, r in recs.items():
        g = gpu[k]
        for name in ("skat_p", "skato_p"):
            b = r[name]
            if not (b > 0):
                continue
            key = "1e%d" % int(np.floor(np.log10(b)))
            e = dec.setdefault(key, {"n": 0, "skat_max_rel": 0.0, "skato_max_rel": 0.0})
            e["n"] += 1
            f = name.replace("_p", "_max_rel")
            e[f] = max(e[f], abs(getattr(g, name) - b) / b)
    worst["by_decade"] = dict(sorted(dec.items(), key=lambda kv: -int(kv[0][2:])))
    return worst


def from_host_rates(eng, blocks, Ms, afs, N, genes=192, window=64):
    """Secondary figures: the same hot path fed from HOST memory through the ModelFitter-style streaming calls (what a
    drop-in rvtests run does): pageable fp64 blocks (rvt_submit_gene, 8 B per genotype over PCIe), int8 hard calls
    (rvt_submit_gene_i8, 1 B), PLINK 2-bit rows (rvt_submit_gene_bed, 1/4 B) and the TEXT of the VCF records
    (rvt_submit_gene_vcf: "0/1<tab>" = 4 B per genotype, split and decoded on the device) and BGEN v1.2 probability
    blocks (rvt_submit_gene_bgen: unphased diploid, 16 bits = 4 B + 1 ploidy byte per genotype); records are taken with
    rvt_collect_ready while the stream runs (no drain) and rvt_collect at the end; every mode streams one untimed window
    of genes first (steady state of a long run)."""
    ks = [k for k in range(len(Ms)) if 40 <= Ms[k] <= 60][:4] or list(range(min(4, len(Ms))))
    host = [np.asfortranarray(blocks[k][:, :N].T.cpu().numpy()) for k in ks]
    hard = [np.rint(h) for h in host]
    out = {}
    for mode in ("fp64", "int8", "bed2bit", "vcf_text", "bgen16", "fp64_registered", "int8_registered",
                 "bed2bit_registered"):
        # *_registered: the same hand-off buffers page-locked once by the caller (rvt_host_register): DMA straight out of
        # them, no staging copy by the CPU
        registered = mode.endswith("_registered")
        mode = mode.replace("_registered", "")
        if mode == "bgen16":
            # probabilities 1 (= 65535) on the called genotype: the dosages are exactly the hard calls
            head = np.array([N], dtype="<u4").tobytes() + np.array([2], dtype="<u2").tobytes() + bytes([2, 2])
            pm = np.full(N, 2, dtype=np.uint8).tobytes() + bytes([0, 16])
            data = []
            for h in hard:
                gi = h.astype(np.int64)
                blks = []
                for j in range(h.shape[1]):
                    v = np.zeros((N, 2), dtype="<u2")
                    v[gi[:, j] == 0, 0] = 65535
                    v[gi[:, j] == 1, 1] = 65535
                    blks.append(head + pm + v.tobytes())
                data.append(blks)
            eng.vcf_set_samples(np.arange(N, dtype=np.int32))
            nbytes = [sum(len(b) for b in d) for d in data]
        elif mode == "vcf_text":
            lut = np.frombuffer(b"0/0\t0/1\t1/1\t", dtype=np.uint8).reshape(3, 4)
            head = b"1\t1000\t.\tA\tG\t50\tPASS\t.\tGT\t"
            eng.vcf_set_samples(np.arange(N, dtype=np.int32))
            data = [eng.prepare_vcf([head + lut[h[:, j].astype(np.int64)].tobytes()[:-1]
                                     for j in range(h.shape[1])]) for h in hard]
            nbytes = [sum(len(ln) for ln in d[1]) for d in data]
        elif mode == "int8":
            data = [np.asfortranarray(h.astype(np.int8)) for h in hard]
        elif mode == "bed2bit":
            data = [eng.pack_bed(h) for h in hard]
        else:
            data = host
        if registered:
            try:
                n_reg = 0
                for a in data:
                    eng.host_register(a)
                    n_reg += 1
            except rvtests_amd.RvtError as e:         # (a host that does not let this user page-lock: report, go on)
                for a in data[:n_reg]:
                    eng.host_unregister(a)
                out[mode + "_registered"] = {"error": str(e)[:200]}
                continue
        done = 0
        t0 = None
        # (the packed hand-offs stream thousands of genes per second: 192 genes would be 40 ms, a third of it the drain)
        n_timed = genes * 8 if mode in ("int8", "bed2bit") else (genes * 4 if mode == "fp64" and not registered else genes)
        # (the text / BGEN modes are the only ones left that fill fp64 blocks on the device — since round 6 the fp64 and int8
        #  hand-offs stay packed — and nobody before them has grown the block pool to its steady state: three windows for them)
        warm = window * (3 if mode in ("vcf_text", "bgen16") else 1)
        for g in range(-warm, n_timed):          # untimed windows first: buffers, block pool and page mappings warm
            if g == 0:
                eng.collect()
                t0 = time.perf_counter()
                done = 0
            i = g % len(ks)
            if mode == "fp64":
                eng.submit_gene(g, data[i], afs[ks[i]])
            elif mode == "int8":
                eng.submit_gene_raw(g, data[i], want_af=False)
            elif mode == "vcf_text":
                eng.submit_gene_vcf(g, data[i], want_af=False)
            elif mode == "bgen16":
                eng.submit_gene_bgen(g, data[i], 2, want_af=False)
            else:
                eng.submit_gene_bed(g, data[i], Ms[ks[i]], want_af=False)
            if (g + 1) % window == 0:
                done += len(eng.collect_ready())
        done += len(eng.collect())
        dt = time.perf_counter() - t0
        if registered:
            for a in data:
                eng.host_unregister(a)
        per_gene = (sum(nbytes) if mode in ("vcf_text", "bgen16") else sum(d.nbytes for d in data)) / len(data)
        out[mode + ("_registered" if registered else "")] = {"gene_sets_per_s": done / dt, "genes": done,
                                                             "host_GBps": per_gene * done / dt / 1e9}
    return out


def resident_bed_rate(eng, blocks, Ms, N, rows_resident=8192, genes=6144, per_call=64, m_lo=20, m_hi=80):
    """Secondary figure: the same four tests on genes of a PLINK .bed matrix RESIDENT in HBM (rvt_bed_alloc / rvt_bed_upload /
    rvt_submit_genes kind 7): a gene is M consecutive rows named by their device address, nothing crosses PCIe, the sufficient
    statistics are formed from the 2-bit rows (gene_suffstat_hcp).  500 000 x 2 000 000 such genotypes are 250 GB: a whole
    exome-scale cohort fits one MI355X.  The matrix here: `rows_resident` rows (the hard calls of four of the batch's genes,
    repeated), genes of M ~ U{m_lo..m_hi} consecutive rows at random offsets; one untimed pass first."""
    ks = [k for k in range(len(Ms)) if 40 <= Ms[k] <= 60][:4] or list(range(min(4, len(Ms))))
    packed = [eng.pack_bed(np.rint(np.asfortranarray(blocks[k][:, :N].T.cpu().numpy()))) for k in ks]
    cb = (N + 3) // 4
    d_bed = eng.bed_alloc(rows_resident)
    r = 0
    while r < rows_resident:
        for pk in packed:
            n = min(pk.shape[0], rows_resident - r)
            if n <= 0:
                break
            eng.bed_upload(d_bed, r, pk[:n])
            r += n
    rng = np.random.default_rng(11)
    gm = rng.integers(m_lo, m_hi + 1, genes)
    first = rng.integers(0, rows_resident - m_hi, genes)
    ptrs = [d_bed + int(f) * cb for f in first]
    out = {}
    try:
        for timed in (False, True):
            n = genes if timed else min(genes, 1024)
            eng.collect()
            t0 = time.perf_counter()
            done = 0
            for g0 in range(0, n, per_call):
                g1 = min(n, g0 + per_call)
                eng.submit_genes_bed_dev(list(range(g0, g1)), ptrs[g0:g1], gm[g0:g1])
                done += len(eng.collect_ready())
            done += len(eng.collect())
            dt = time.perf_counter() - t0
        # the single-variant score test over the same rows (rvt_score_bed_dev; median of five calls)
        eng.score_bed_dev(d_bed, rows_resident, want_counts=False)
        ts = []
        for _ in range(5):
            t1 = time.perf_counter()
            eng.score_bed_dev(d_bed, rows_resident)
            ts.append(time.perf_counter() - t1)
        score_rate = rows_resident / float(np.median(ts))
        out = {"gene_sets_per_s": done / dt, "genes": done, "rows_resident": rows_resident,
               "score_test_variants_per_s": score_rate,
               "resident_GB": rows_resident * cb / 1e9, "mean_M": float(gm.mean()),
               "workload": "N=%d, genes = M~U{%d..%d} consecutive rows of a device-resident .bed matrix, all four tests; "
                           "caller: Python, %d genes per rvt_submit_genes call" % (N, m_lo, m_hi, per_call)}
    finally:
        eng.bed_free(d_bed)
    return out


def spawn_ranks(n):
    """`--gpus n` without a launcher: start n ranks as a child torch.distributed.run BEFORE this process touches the
    GPU, pass its output through and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    if _CPU_WORKER:
        cpu_worker(sys.argv[2])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--samples", type=int, default=500000)
    ap.add_argument("--genes", type=int, default=512, help="genes per step per GPU (512 x ~200 MB = 102 GB resident)")
    ap.add_argument("--m-lo", type=int, default=20)
    ap.add_argument("--m-hi", type=int, default=80)
    ap.add_argument("--missing-frac", type=float, default=0.05,
                    help="share of the genes with missing genotypes (0.1 %% of their calls, imputed to the column mean); "
                         "SURVEY config 3: 0.05")
    ap.add_argument("--dosage", action="store_true",
                    help="every block holds dosages with three decimals (imputed data, VCF DS fields); the engine is told "
                         "so (rvt_set_content_hint, rvt_set_dosage_lattice) as an adapter reading --dosage input would")
    ap.add_argument("--dosage-lattice", type=int, default=1000,
                    help="with --dosage: the lattice denominator stated to the engine (1000 = three decimals: the int8 "
                         "lattice kernel); 0 = not stated: the general fp64 kernel")
    ap.add_argument("--dosage-float", action="store_true",
                    help="with --dosage: float-precision dosages as an 8-bit BGEN file gives them (multiples of 2^-31), no lattice "
                         "stated; the engine is told so (rvt_set_dosage_float) and tries the float-digit int8 kernel for M <= 64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-literal-baseline", action="store_true", help="skip the literal N x N SKAT timing (~20 s of CPU)")
    ap.add_argument("--no-from-host", action="store_true", help="skip the from-host (PCIe-inclusive) secondary rates")
    ap.add_argument("--cpu-genes", type=int, default=24,
                    help="genes of the batch run through the CPU oracle after the timed region (baseline + parity)")
    ap.add_argument("--tests", type=int, default=rvtests_amd.TEST_ALL)
    ap.add_argument("--trait", choices=["qt", "binary"], default="qt",
                    help="qt = BASELINE configs[2] (default); binary = configs[3]-style logistic null")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))       # nothing has touched the GPU yet
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # one rank per GPU; RVT_BENCH_BACKEND=gloo lets two ranks share GPU 0 to exercise the N > 1 path on a 1-GPU box
    backend = os.environ.get("RVT_BENCH_BACKEND", "nccl")
    gpu_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    N = args.samples
    # this rank on the NUMA node its device hangs on (rvt_pin_to_device_node: what a main() built on the library calls first; the
    # staging threads inherit the mask).  The timed region does not depend on it; the from-host figures do.  RVT_NO_PIN=1: off
    pinned_node = -1
    if not os.environ.get("RVT_NO_PIN"):
        try:
            L0 = rvtests_amd.load_library()
            L0.rvt_pin_to_device_node.restype = ctypes.c_int
            L0.rvt_pin_to_device_node.argtypes = [ctypes.c_int]
            pinned_node = int(L0.rvt_pin_to_device_node(int(gpu_index)))
        except Exception:
            pinned_node = -1
    eng = rvtests_amd.Engine(gpu_index)
    ld = eng.padded_ld(N)

    # ---- null model: phenotype and covariates from rank 0 (collective C1: one broadcast), fitted and installed on
    #      every rank by the engine itself (rvt_fit_null: OLS / IRLS on the device) -----------------------------
    d = 3
    binary = args.trait == "binary"
    # ---- this rank's shard of genes, resident in HBM -----------------------------------------------------------
    blocks, Ms, afs = make_genes(dev, N, ld, args.genes, 20260002 + 1000 * rank, args.m_lo, args.m_hi,
                                 args.missing_frac, args.dosage, args.dosage_float)
    if args.dosage_float:
        args.dosage_lattice = 0
    if args.dosage:
        eng.set_content_hint(0)
        eng.set_dosage_lattice(args.dosage_lattice)
        if args.dosage_float:
            eng.set_dosage_float(True)
    pack = torch.empty((N, d + 1), dtype=torch.float64, device=dev)
    if rank == 0:
        X, y = make_phenotype(dev, N, 20260002, binary, causal_effect(blocks, N, binary))
        pack[:, :d] = X
        pack[:, d] = y
    if world > 1:
        dist.broadcast(pack, 0)
    Xh = np.asfortranarray(pack[:, :d].cpu().numpy())
    yh = pack[:, d].cpu().numpy().copy()
    eng.fit_null(rvtests_amd.TRAIT_BINARY if binary else rvtests_amd.TRAIT_QUANTITATIVE, Xh, yh)

    torch.cuda.synchronize()
    # (the blocks are torch allocations the engine knows nothing about: no classification pass, nothing registered —
    # the hard-call kernel tests what it loads, in the timed region)
    # pre-packed batches over the same resident blocks: the engine keeps up to four batches in flight (one HIP
    # stream each), so the latency-bound tail of step i overlaps the bandwidth-bound head of step i+1
    # batches in flight: RVT_MAX_INFLIGHT by default (RVT_BENCH_INFLIGHT lowers it for experiments)
    NSLOT = max(1, min(rvtests_amd.MAX_INFLIGHT, int(os.environ.get("RVT_BENCH_INFLIGHT", rvtests_amd.MAX_INFLIGHT))))
    batches = [eng.prepare([b.data_ptr() for b in blocks], Ms, afs, tests=args.tests) for _ in range(NSLOT)]
    batch = batches[0]
    eng.reserve(Ms)     # every pipeline slot's workspace up front: no allocation inside a step, warm-up or timed

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from rvtests_amd import shard
    counts = [args.genes] * world
    inflight = []

    last_gather = {"n": None, "ordered": None, "seconds": 0.0, "calls": 0}

    def retire(b):
        """step finished on this rank: C2 = gather its per-gene records on rank 0 (gene order)"""
        if world > 1:
            rec = shard.records_from_results(b["out"])
            rec[:, 0] += rank * args.genes          # gene ids are rank-local: make them global for the ordered merge
            tg = time.perf_counter()
            allr = shard.gather_records(dist, rec, counts, device=dev if backend == "nccl" else None, dst=0)
            last_gather["seconds"] += time.perf_counter() - tg
            last_gather["calls"] += 1
            if rank == 0:
                last_gather["n"] = int(allr.shape[0])
                last_gather["ordered"] = bool(np.array_equal(allr[:, 0], np.arange(world * args.genes)))

    def run_steps(k):
        for i in range(k):
            b = batches[i % NSLOT]
            if len(inflight) == NSLOT:      # launching into a busy slot first finishes the oldest batch
                eng.wait_oldest()
                retire(inflight.pop(0))
            eng.launch(b)
            inflight.append(b)
        eng.sync()
        while inflight:
            retire(inflight.pop(0))

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else None)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    barrier()
    tw = time.perf_counter()
    run_steps(args.warmup)
    barrier()
    warm_ms = max_over_ranks(time.perf_counter() - tw) * 1e3 / max(args.warmup, 1)
    # The timed region is EXACTLY args.steps steps between barriers, measured once.
    eng.set_profiling(True)
    eng.timing(reset=True)
    t0 = time.perf_counter()
    run_steps(args.steps)
    own_elapsed = time.perf_counter() - t0       # (this rank's own steps, before the barrier: per-rank rate below)
    barrier()
    elapsed = time.perf_counter() - t0
    tm = eng.timing(reset=True)
    eng.set_profiling(False)
    elapsed = max_over_ranks(elapsed)
    # every rank's own rate and C2 time, for rank 0's line: the first real multi-GPU run is self-checking
    per_rank = None
    if world > 1:
        mine = torch.tensor([args.genes * args.steps / own_elapsed, last_gather["seconds"] / max(last_gather["calls"], 1)],
                            dtype=torch.float64, device=dev if backend == "nccl" else None)
        allm = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        per_rank = [[float(t[0]), float(t[1])] for t in allm]

    if rank == 0:
        total_genes = world * args.genes * args.steps
        value = total_genes / elapsed
        out0 = batches[(args.steps - 1) % NSLOT]["out"] if args.steps else batch["out"]
        T = args.tests
        # (genes whose requested kernel tests all came back with a p-value; the flags of the line below name what was asked)
        ok = sum(1 for r in out0 if (r.skat_ok or not T & rvtests_amd.TEST_SKAT) and (r.skato_ok or not T & rvtests_amd.TEST_SKATO))
        kern = ",".join(n for n, b in (("skat[nPerm=0]", rvtests_amd.TEST_SKAT), ("skato", rvtests_amd.TEST_SKATO)) if T & b)
        burd = ",".join(n for n, b in (("cmc", rvtests_amd.TEST_CMC), ("zeggini", rvtests_amd.TEST_ZEGGINI)) if T & b)
        flags = " ".join(x for x in ("--kernel " + kern if kern else "", "--burden " + burd if burd else "") if x)
        names = "+".join(n for n, b in (("SKAT", rvtests_amd.TEST_SKAT), ("SKAT-O", rvtests_amd.TEST_SKATO),
                                        ("CMC", rvtests_amd.TEST_CMC), ("Zeggini", rvtests_amd.TEST_ZEGGINI)) if T & b)
        cfg_index = 3 if binary else (1 if (N == 50000 and T == rvtests_amd.TEST_SKAT) else 2)
        # roofline of the sufficient-statistics kernel (the contraction over N): algorithmic bytes / measured time.
        # The dominant kernel is gene_suffstat_hc (hard-call blocks: ~95 % of the genes and bytes of this workload);
        # the few blocks with imputed means take the general fp64 kernel gene_suffstat_mfma, which runs BESIDE it on
        # a second stream — its numbers are reported separately (an overlapped kernel's own duration is not chip time).
        if tm.n_suffstat_hc_launches > 0:
            # (binary trait: the weighted kernels — gene_suffstat_hcx, or gene_suffstat_hcw with RVT_HCX=0 / a null model it does not take)
            k2_name = eng.hardcall_kernel() or ("gene_suffstat_hcw" if binary else "gene_suffstat_hc")
            if args.dosage:
                k2_name = "gene_suffstat_lat"                                 # (dosages on the stated decimal lattice)
            if args.dosage_float:
                k2_name = "gene_suffstat_fdx"                                 # (float-precision dosages, M <= 64)
            n_l, ms_l, by_l = tm.n_suffstat_hc_launches, tm.ms_suffstat_hc, tm.alg_bytes_hc
        else:
            k2_name = "gene_suffstat_mfma"
            n_l, ms_l, by_l = tm.n_suffstat_launches, tm.ms_suffstat, tm.alg_bytes
        ms_k2 = ms_l / max(n_l, 1)
        bytes_per_launch = by_l / max(n_l, 1)
        achieved = bytes_per_launch / (ms_k2 * 1e-3) / 1e9 if ms_k2 > 0 else 0.0
        n_g, ms_g, by_g = (tm.n_suffstat_launches - tm.n_suffstat_hc_launches, tm.ms_suffstat - tm.ms_suffstat_hc,
                           tm.alg_bytes - tm.alg_bytes_hc)
        tot_ms = tm.ms_suffstat + tm.ms_burden + tm.ms_stats + tm.ms_pvalue
        # HBM traffic per launch from the committed PMC passes of this same workload (tools/collect_profiles.sh,
        # tools/pmc_traffic.py: FETCH_SIZE x2 on gfx950 + WRITE_SIZE); null when the workload differs
        traffic = None
        key = "N=%d,genes=%d,m=%d..%d,seed=20260002,tests=%d" % (N, args.genes, args.m_lo, args.m_hi, args.tests)
        if binary:
            key += ",binary"
        if args.dosage:                                  # (the dosage workload has PMC passes of its own)
            key += ",dosage,lattice=%d" % args.dosage_lattice
        if args.dosage_float:
            key += ",float"
        # (the newest committed PMC summary of exactly this workload: profiles/r*_pmc_traffic*.json, tools/pmc_traffic.py)
        import glob
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        traffic_src = None
        for pmc_path in sorted(glob.glob(os.path.join(pdir, "r*_pmc_traffic*.json")), reverse=True):
            try:
                pmc = json.load(open(pmc_path))
            except Exception:
                continue
            k2 = pmc.get("kernels", {}).get("suffstat_lat" if k2_name == "gene_suffstat_lat" else
                                            ("suffstat_hc" if k2_name.startswith("gene_suffstat_hc") else "suffstat"))
            if pmc.get("workload") == key and k2 and pmc.get("kernel_name", k2_name) == k2_name:
                traffic = k2["hbm_bytes_per_step"] / k2["launches_per_step"]
                traffic_src = os.path.basename(pmc_path)
                break
        line = {
            "metric": "gene-sets/sec (%s, analytic p-values)" % names,
            "value": value, "unit": "gene-sets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: N=%d, genes/step/GPU=%d, M~U{%d..%d}, %s trait, "
                                   "d=3, %s; genes resident in HBM as "
                                   "fp64 column-major blocks%s" % (cfg_index, N, args.genes, args.m_lo,
                                                                   args.m_hi, "binary" if binary else "quantitative", flags,
                                                                   (" of float-precision DOSAGES (8-bit BGEN values; stated to the engine)" if args.dosage_float else
                                                                    " of DOSAGES with three decimals (no hard-call block; lattice stated: %d)"
                                                                    % args.dosage_lattice) if args.dosage else ""),
                       "N": N, "genes_per_step_per_gpu": args.genes, "mean_M": float(np.mean(Ms)),
                       "parallelism": "gene-sharded x%d" % world, "genes_ok": ok,
                       "hard_call_genes_per_step": (tm.genes_hard_call - tm.genes_handed_back) / max(args.steps, 1),
                       "genes_handed_back_per_step": tm.genes_handed_back / max(args.steps, 1),
                       "genes_with_imputed_columns": 0.0 if args.dosage else args.missing_frac},
            "roofline": {"kernel": k2_name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": ms_k2, "launches": int(n_l),
                         "general_fp64_kernel": {"kernel": "gene_suffstat_mfma", "launches": int(n_g),
                                                 "avg_launch_ms": ms_g / max(n_g, 1),
                                                 "achieved": (by_g / max(ms_g, 1e-9)) / 1e6 if n_g else None,
                                                 "note": "runs concurrently with the hard-call launches"},
                         "all_suffstat_algorithmic_GBps_over_summed_durations":
                             (tm.alg_bytes / max(tm.ms_suffstat, 1e-9)) / 1e6},
            "kernel_time_share": {"suffstat_mfma": tm.ms_suffstat / tot_ms, "burden": tm.ms_burden / tot_ms,
                                  "gene_stats": tm.ms_stats / tot_ms, "gene_pvalue": tm.ms_pvalue / tot_ms,
                                  "device_ms_per_step": tot_ms / args.steps},
            "pinned_numa_node": pinned_node,
            "davies_terms_per_gene": float(np.mean([r.davies_terms for r in out0])),
            "warmup_ms_per_step": warm_ms,
        }
        if world == 1 and not args.no_from_host and not binary:
            try:
                line["from_host"] = from_host_rates(eng, blocks, Ms, afs, N)
            except Exception as e:                        # (secondary figures must never cost the line)
                line["from_host"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_from_host and not binary and not args.dosage and args.tests == rvtests_amd.TEST_ALL:
            try:
                line["resident_bed"] = resident_bed_rate(eng, blocks, Ms, N)
            except Exception as e:
                line["resident_bed"] = {"error": repr(e)[:300]}
        if world > 1:
            line["ranks_seen"] = int(dist.get_world_size())
            line["devices_seen_by_rank0"] = int(torch.cuda.device_count())
            line["per_rank_gene_sets_per_s"] = [r[0] for r in per_rank]
            line["c2_gather_ms_per_step_by_rank"] = [1e3 * r[1] for r in per_rank]
            line["gathered_records_last_step"] = last_gather["n"]
            line["gathered_ids_in_order"] = last_gather["ordered"]
        if world == 1 and not args.no_cpu_baseline and args.cpu_genes > 0:
            # ---- CPU baseline + parity on a sample of the batch's genes (after the timed region) ------------------
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 1
            try:
                import psutil
                by_mem = int(psutil.virtual_memory().available / (3 << 30))
            except Exception:
                by_mem = 4
            workers = max(1, min(ncpu, by_mem, args.cpu_genes))
            n_sample = min(args.cpu_genes, args.genes)
            # spread over the batch's widths: every (genes / n_sample)-th gene in order of M
            by_m = np.argsort(np.array(Ms), kind="stable")
            pick = [int(by_m[int(round(i * (len(by_m) - 1) / max(n_sample - 1, 1)))]) for i in range(n_sample)]
            # ... of which the first half are the causal genes (small p-values), the rest null genes of every width
            n_c = min(N_CAUSAL, n_sample // 2, args.genes)
            pick = sorted(set(list(range(n_c)) + pick[:max(n_sample - n_c, 1)]))
            k1 = int(np.argmin(np.abs(np.array(Ms) - 50)))           # the 1-thread gene: mean width
            host = {k: (np.asfortranarray(blocks[k][:, :N].T.cpu().numpy()), afs[k]) for k in set(pick) | {k1}}
            one, wall1 = cpu_oracle_pool({k1: host[k1]}, Xh, yh, 1 if binary else 0, 1)
            if one:
                t1 = one[k1]["seconds"]
                line["cpu_baseline"] = {"value": 1.0 / t1, "unit": "gene-sets/s", "cores": 1, "kind": "port",
                                        "sample": "1 gene of the batch (M=%d, N=%d) alone on the host: oracle folded SKAT "
                                                  "+ literal SKAT-O + CMC + Zeggini, g++ -O2 -msse2 (the reference's "
                                                  "release flags), %.1f s (null fit excluded)" % (Ms[k1], N, t1)}
                # the same sources built -O3 -march=native (what a tuned CPU build of the same loops gives)
                nat = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "liboracle_native.so")
                if os.path.exists(nat):
                    one_n, _ = cpu_oracle_pool({k1: host[k1]}, Xh, yh, 1 if binary else 0, 1, library=nat)
                    if one_n:
                        line["cpu_baseline"]["native_build"] = {
                            "value": 1.0 / one_n[k1]["seconds"], "unit": "gene-sets/s", "cores": 1,
                            "flags": "g++ -O3 -march=native", "seconds": one_n[k1]["seconds"]}
                # B-lit-SKAT: the reference's own formulation — Skat.cpp:57-76 forms the N x N P0 in float —
                # cannot run at this N (4 N^2 bytes = 1 TB); timed at N = 4 000 / 8 000 / 16 000 on one thread, its fitted
                # power law gives what a gene would cost here
                if not args.no_literal_baseline:
                    try:
                        line["cpu_baseline_literal_skat"] = literal_skat_baseline(N)
                    except Exception as e:
                        line["cpu_baseline_literal_skat"] = {"error": repr(e)[:200]}
            # every core the affinity mask gives, bounded by memory and by 48: the sampled genes (parity) plus replicas of the
            # mean-width gene until there is one gene per process — what the CPU port delivers per host with genes dealt to cores
            W = max(len(pick), min(ncpu, by_mem, 48))
            pool_genes = {k: host[k] for k in pick}
            for i in range(W - len(pick)):
                pool_genes[-(i + 1)] = host[k1]
            recs, wall = cpu_oracle_pool(pool_genes, Xh, yh, 1 if binary else 0, W)
            if recs:
                n_all = len(recs)
                recs = {k: r for k, r in recs.items() if k >= 0}
                line["cpu_baseline_multiproc"] = {
                    "value": n_all / wall, "unit": "gene-sets/s", "cores": W, "host_cores": ncpu, "kind": "port",
                    "sample": "%d genes (the %d sampled genes of the batch, M %d..%d, + %d replicas of the M=%d gene) on %d "
                              "single-thread processes at once, one gene each, %.1f s wall (process start-up and one null fit "
                              "per process included); the affinity mask has %d cores, memory allows %d processes"
                              % (n_all, len(pick), min(Ms[k] for k in pick), max(Ms[k] for k in pick), W - len(pick), Ms[k1],
                                 W, wall, ncpu, by_mem)}
                if one:
                    recs.setdefault(k1, one[k1])
                line["parity"] = parity_summary(recs, out0)
                line["pvalue_max_abs_diff"] = line["parity"]["pvalue_max_abs_diff"]
                line["q_max_rel_diff"] = line["parity"]["q_max_rel_diff"]
        if world == 1 and not args.no_from_host and not binary:
            # the same hand-off from a C++ caller (tools/host_feed_bench: one thread, rvt_submit_gene_* per gene — what the
            # drop-in adapters do), as a child process AFTER this process has closed its engine: two contexts on one device
            # oversubscribe its hardware queues (every context creates its streams with queues of their own) — the
            # figures of rounds 3-4, taken beside the parent's idle context, were 30-40 % low for that reason
            tool = os.path.join(ROOT, "tools", "host_feed_bench")
            if os.path.exists(tool):
                import subprocess
                try:
                    eng.close()
                    del blocks
                    torch.cuda.empty_cache()
                    pr = subprocess.run([tool, "--samples", str(N), "--m", "50", "--genes", "1536", "--modes", "int8,bed"],
                                        capture_output=True, text=True, timeout=240, env=dict(os.environ, RVT_TRACE_SUBMIT="1"))
                    recs = [json.loads(ln) for ln in pr.stdout.splitlines() if ln.startswith("{")]
                    line["from_host_cpp"] = [r for r in recs if "diag" not in r and "link_settle" not in r]
                    line["from_host_cpp_link_settle"] = next((r["link_settle"] for r in recs if "link_settle" in r), None)
                    # the engine's own account of the caller's time per gene (RVT_TRACE_SUBMIT: staging copy incl. the wait for a
                    # free pinned chunk, consolidation launches, batch launches), both modes together
                    line["from_host_cpp_submit_trace"] = [ln.strip() for ln in pr.stderr.splitlines() if "submit trace" in ln][:4]
                    # what this box gives the hand-off to work with (rvt_host_diagnose: threads, NUMA nodes, memcpy / staging /
                    # DMA rates measured in place): a from-host figure far below another box's is explained by these or by
                    # nothing the engine controls
                    line["from_host_cpp_diag"] = next((r["diag"] for r in recs if "diag" in r), None)
                except Exception as e:
                    line["from_host_cpp"] = {"error": repr(e)[:300]}
            # the DROP-IN itself at the size of the headline: the C++ adapters end to end (ModelManager::create, fit() /
            # writeOutput() per gene and model as src/Main.cpp:1221-1254 drives them, GpuBroker's deferred gene-ordered rows,
            # writeFootnote()) over 1 536 synthetic genes of N samples, M ~ U{20..80}, SKAT + SKAT-O + CMC + Zeggini, handed
            # over as dc->getGenotype() holds them (fp64, 8 bytes per genotype: the only layout a stock rvtest has) and as
            # PLINK 2-bit rows before consolidation; the two runs print the same .assoc text (rows_sha256)
            drv = os.path.join(ROOT, "rvtests_amd", "csrc", "host", "host_driver")
            if os.path.exists(drv):
                import subprocess
                di = {}
                for mode in ("fp64", "bed"):
                    try:
                        pr = subprocess.run([drv, "--synthetic", str(N), "1536", "20", "80", mode, "skat[nPerm=0],skato", "cmc,zeggini",
                                             "--pool", "16"], capture_output=True, text=True, timeout=300)
                        rec = [json.loads(ln) for ln in pr.stdout.splitlines() if ln.startswith("{")]
                        di[mode] = rec[0] if rec else {"error": (pr.stderr or "no output")[-300:]}
                    except Exception as e:
                        di[mode] = {"error": repr(e)[:300]}
                di["rows_identical"] = bool(di.get("fp64", {}).get("rows_sha256")) and \
                    di["fp64"].get("rows_sha256") == di.get("bed", {}).get("rows_sha256")
                # `--meta cov` the same way: MetaCovTest::fit() per variant with its fp64 column (the reference's single-variant
                # loop, src/Main.cpp:1010-1078), windows of 200 markers, rows formatted and hashed on the host
                # (meta_cov_imputed: 1 % of every column's calls missing and imputed to the column mean, as real matrices are —
                #  the window's band then is four MXFP4 products instead of one)
                for key, extra in (("meta_cov", []), ("meta_cov_imputed", ["--missing", "0.01"])):
                    try:
                        pr = subprocess.run([drv, "--synthetic-meta", str(N), "8000", "200"] + extra, capture_output=True, text=True,
                                            timeout=300)
                        rec = [json.loads(ln) for ln in pr.stdout.splitlines() if ln.startswith("{")]
                        di[key] = rec[0] if rec else {"error": (pr.stderr or "no output")[-300:]}
                    except Exception as e:
                        di[key] = {"error": repr(e)[:300]}
                line["drop_in"] = di
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
