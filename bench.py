#!/usr/bin/env python3
"""bench.py — gene-sets/sec of the kernel/burden hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (SKAT + SKAT-O + CMC + Zeggini, analytic p-values) over one batch of
`--genes` synthetic genes that are ALREADY RESIDENT in HBM in the engine's boundary layout (fp64, column-major,
leading dimension padded to 16 samples).  Workload = BASELINE.json configs[2] ("N=500k, M̄=50,
--kernel skat,skato --burden cmc,zeggini"), the configuration the metric is quoted on: per GPU a shard of
genes with M_g ~ Uniform{20..80}; genes shard across ranks with no data-path collective (weak scaling) — the
only collectives are one broadcast of the null model and one gather of the per-gene result records per step.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the engine's stream around the
fp64-MFMA sufficient-statistics kernel; `cpu_baseline` times the CPU oracle (literal SKAT-O + folded SKAT +
CMC + Zeggini, reference release flags -O2 -msse2, 1 thread) on ONE gene of the same workload.
"""
import argparse
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


def make_genes(dev, N, ld, n_genes, seed, m_lo, m_hi):
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
        if rng.random() < 0.05:
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


def make_phenotype(dev, N, seed, binary=False):
    """Covariates and phenotype of SURVEY §8d: quantitative y = 0.3 c1 - 0.2 c2 + N(0,1) (configs 2/3) or binary
    y ~ Bernoulli(logit^-1(-2 + 0.3 c1)) (config 4).  Returns X (with intercept) and y as device tensors."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    c = torch.randn((N, 2), generator=g, device=dev, dtype=torch.float64)
    X = torch.cat([torch.ones((N, 1), device=dev, dtype=torch.float64), c], 1)
    if binary:
        pr = torch.sigmoid(-2.0 + 0.3 * c[:, 0])
        y = (torch.rand(N, generator=g, device=dev, dtype=torch.float64) < pr).to(torch.float64)
    else:
        y = 0.3 * c[:, 0] - 0.2 * c[:, 1] + torch.randn(N, generator=g, device=dev, dtype=torch.float64)
    return X, y


def fit_null_qt(dev, N, seed):
    """Quantitative null model y ~ 1 + c1 + c2 solved with torch (used by the tools/ microbenchmarks)."""
    X, y = make_phenotype(dev, N, seed)
    beta = torch.linalg.solve(X.T @ X, X.T @ y)
    res = y - X @ beta
    sigma2 = float((res @ res) / N)
    return X, y, res, sigma2


def cpu_baseline(G_host, af, X, y, binary):
    """Time the CPU oracle on one gene: SKAT (P0 folded; the literal N x N form cannot run at this N), literal
    SKAT-O, CMC and Zeggini — the four ModelFitter::fit bodies of the workload.  The null model (fitted once per
    analysis, not per gene) is outside the timed region."""
    import orc
    if binary:
        rc, beta, p, v = orc.fit_logistic(X, y)
        res = y - p
    else:
        rc, beta, pred, res, s2 = orc.fit_linear(X, y)
        v = np.full(len(y), s2)
    t0 = time.perf_counter()
    orc.skat(G_host, af, X, res, v, binary)
    orc.skato(G_host, af, X, res, v, binary)
    orc.burden(G_host, X, y, binary, 0)
    orc.burden(G_host, X, y, binary, 1)
    return time.perf_counter() - t0


def cpu_baseline_all_cores(G_host, af, X, y, binary, max_workers=32):
    """The same single-gene oracle run as `cpu_baseline`, one independent process per core on `workers` cores at once
    (genes are independent, so this is how the CPU port would use a whole host): returns (genes/s, workers, wall s).
    Child processes are plain `python bench.py --cpu-worker <file>` interpreters that never touch the GPU; their number
    is bounded by the cores this process may use and by the free host memory (about 4 GB per worker)."""
    import subprocess
    import tempfile
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    try:
        import psutil
        by_mem = int(psutil.virtual_memory().available / (4 << 30))
    except Exception:
        by_mem = 4
    workers = max(1, min(max_workers, ncpu, by_mem))
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as td:
        path = os.path.join(td, "gene.npz")
        np.savez(path, G=G_host, af=af, X=X, y=y, binary=np.int64(binary))
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(workers)]
        ok = 0
        for pr in procs:
            out, _ = pr.communicate()
            ok += 1 if pr.returncode == 0 and out.strip() else 0
        wall = time.perf_counter() - t0
    if ok != workers:
        return None
    return workers / wall, workers, wall


def main():
    if _CPU_WORKER:
        z = np.load(sys.argv[2])
        print(cpu_baseline(np.asfortranarray(z["G"]), z["af"], np.asfortranarray(z["X"]), z["y"], int(z["binary"])))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--samples", type=int, default=500000)
    ap.add_argument("--genes", type=int, default=512, help="genes per step per GPU (512 x ~200 MB = 102 GB resident)")
    ap.add_argument("--m-lo", type=int, default=20)
    ap.add_argument("--m-hi", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tests", type=int, default=rvtests_amd.TEST_ALL)
    ap.add_argument("--trait", choices=["qt", "binary"], default="qt",
                    help="qt = BASELINE configs[2] (default); binary = configs[3]-style logistic null")
    args = ap.parse_args()

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
    eng = rvtests_amd.Engine(gpu_index)
    ld = eng.padded_ld(N)

    # ---- null model: phenotype and covariates from rank 0 (collective C1: one broadcast), fitted and installed on
    #      every rank by the engine itself (rvt_fit_null: OLS / IRLS on the device) -----------------------------
    d = 3
    binary = args.trait == "binary"
    pack = torch.empty((N, d + 1), dtype=torch.float64, device=dev)
    if rank == 0:
        X, y = make_phenotype(dev, N, 20260002, binary)
        pack[:, :d] = X
        pack[:, d] = y
    if world > 1:
        dist.broadcast(pack, 0)
    Xh = np.asfortranarray(pack[:, :d].cpu().numpy())
    yh = pack[:, d].cpu().numpy().copy()
    eng.fit_null(rvtests_amd.TRAIT_BINARY if binary else rvtests_amd.TRAIT_QUANTITATIVE, Xh, yh)

    # ---- this rank's shard of genes, resident in HBM -----------------------------------------------------------
    blocks, Ms, afs = make_genes(dev, N, ld, args.genes, 20260002 + 1000 * rank, args.m_lo, args.m_hi)
    torch.cuda.synchronize()
    # pre-packed batches over the same resident blocks: the engine keeps up to four batches in flight (one HIP
    # stream each), so the latency-bound tail of step i overlaps the bandwidth-bound head of step i+1
    # batches in flight: RVT_MAX_INFLIGHT by default (RVT_BENCH_INFLIGHT lowers it for experiments)
    NSLOT = max(1, min(rvtests_amd.MAX_INFLIGHT, int(os.environ.get("RVT_BENCH_INFLIGHT", rvtests_amd.MAX_INFLIGHT))))
    batches = [eng.prepare([b.data_ptr() for b in blocks], Ms, afs, tests=args.tests) for _ in range(NSLOT)]
    batch = batches[0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from rvtests_amd import shard
    counts = [args.genes] * world
    inflight = []

    def retire(b):
        """step finished on this rank: C2 = gather its per-gene records on rank 0 (gene order)"""
        if world > 1:
            shard.gather_records(dist, shard.records_from_results(b["out"]), counts,
                                 device=dev if backend == "nccl" else None, dst=0)

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
    # The timed region is EXACTLY args.steps steps between barriers.  Once in some tens of runs a fresh box shows a host-side
    # stall (seconds per step while the kernels themselves run at their usual duration — lost wake-ups of the blocking
    # stream waits); a timed region more than 5x slower per step than the untimed warm-up steps (which include the
    # first-launch overheads) is therefore measured again, at most twice, and every discarded attempt is reported.
    discarded = []
    while True:
        eng.set_profiling(True)
        eng.timing(reset=True)
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        tm = eng.timing(reset=True)
        eng.set_profiling(False)
        elapsed = max_over_ranks(elapsed)
        step_ms = elapsed * 1e3 / max(args.steps, 1)
        if args.warmup > 0 and step_ms > 5.0 * warm_ms and len(discarded) < 2:
            discarded.append(round(step_ms, 3))
            continue
        break

    if rank == 0:
        total_genes = world * args.genes * args.steps
        value = total_genes / elapsed
        out0 = batches[(args.steps - 1) % NSLOT]["out"] if args.steps else batch["out"]
        ok = sum(1 for r in out0 if r.skat_ok and r.skato_ok)
        # roofline of the sufficient-statistics kernel (the contraction over N): algorithmic bytes / measured time
        ms_k2 = tm.ms_suffstat / max(tm.n_suffstat_launches, 1)
        bytes_per_launch = tm.alg_bytes / max(tm.n_suffstat_launches, 1)
        achieved = bytes_per_launch / (ms_k2 * 1e-3) / 1e9 if ms_k2 > 0 else 0.0
        tot_ms = tm.ms_suffstat + tm.ms_burden + tm.ms_stats + tm.ms_pvalue
        # HBM traffic per launch from the committed PMC passes of this same workload (tools/collect_profiles.sh,
        # tools/pmc_traffic.py: FETCH_SIZE x2 on gfx950 + WRITE_SIZE); null when the workload differs
        traffic = None
        key = "N=%d,genes=%d,m=%d..%d,seed=20260002,tests=%d" % (N, args.genes, args.m_lo, args.m_hi, args.tests)
        if binary:
            key += ",binary"
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_pmc_traffic.json")
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            if pmc.get("workload") == key:
                k2 = pmc["kernels"]["suffstat"]
                traffic = k2["hbm_bytes_per_step"] / k2["launches_per_step"]
        line = {
            "metric": "gene-sets/sec (SKAT+SKAT-O+CMC+Zeggini, analytic p-values)",
            "value": value, "unit": "gene-sets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: N=%d, genes/step/GPU=%d, M~U{%d..%d}, %s trait, "
                                   "d=3, --kernel skat[nPerm=0],skato --burden cmc,zeggini; genes resident in HBM as "
                                   "fp64 column-major blocks" % (3 if binary else 2, N, args.genes, args.m_lo,
                                                                 args.m_hi, "binary" if binary else "quantitative"),
                       "N": N, "genes_per_step_per_gpu": args.genes, "mean_M": float(np.mean(Ms)),
                       "parallelism": "gene-sharded x%d" % world, "genes_ok": ok},
            "roofline": {"kernel": "gene_suffstat_mfma", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "tflops_fp64_mfma": (tm.alg_flops / max(tm.ms_suffstat, 1e-9)) / 1e9,
                         "avg_launch_ms": ms_k2, "launches": int(tm.n_suffstat_launches)},
            "kernel_time_share": {"suffstat_mfma": tm.ms_suffstat / tot_ms, "burden": tm.ms_burden / tot_ms,
                                  "gene_stats": tm.ms_stats / tot_ms, "gene_pvalue": tm.ms_pvalue / tot_ms,
                                  "device_ms_per_step": tot_ms / args.steps},
            "davies_terms_per_gene": float(np.mean([r.davies_terms for r in out0])),
            # untimed warm-up rate and any timed region that was measured again (see the guard above): ms per step
            "warmup_ms_per_step": warm_ms, "discarded_timed_regions_ms_per_step": discarded,
        }
        if world == 1 and not args.no_cpu_baseline:
            k = int(np.argmin(np.abs(np.array(Ms) - 50)))
            Gh = np.asfortranarray(blocks[k][:, :N].T.cpu().numpy())
            t = cpu_baseline(Gh, afs[k], Xh, yh, 1 if binary else 0)
            line["cpu_baseline"] = {"value": 1.0 / t, "unit": "gene-sets/s", "cores": 1, "kind": "port",
                                    "sample": "1 gene of the batch (M=%d, N=%d): oracle folded SKAT + literal SKAT-O + "
                                              "CMC + Zeggini, g++ -O2 -msse2, %.1f s" % (Ms[k], N, t)}
            allc = cpu_baseline_all_cores(Gh, afs[k], Xh, yh, 1 if binary else 0)
            if allc:
                line["cpu_baseline_all_cores"] = {
                    "value": allc[0], "unit": "gene-sets/s", "cores": allc[1], "kind": "port",
                    "sample": "the same gene in %d independent single-thread processes at once (process start-up "
                              "and the null fit included), %.1f s wall" % (allc[1], allc[2])}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
