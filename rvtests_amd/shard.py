"""Gene sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

Genes/windows are independent units that share only the null model, so the data path needs no
collective: rank r processes its shard and the only exchanges are
  C1  one broadcast of the null model (8*N*(d+2) bytes) from rank 0, and
  C2  one gather of fixed-size per-gene result records to rank 0, re-ordered to submission order because the
      reference's output files are gene-ordered (src/Main.cpp:1249-1253).
This module is backend-agnostic: backend "nccl" (= RCCL) on the GPUs, "gloo" in the CPU tests.
"""
import numpy as np

RECORD_FIELDS = ("gene_id", "status", "n_poly", "skat_Q", "skat_p", "skato_Q", "skato_rho", "skato_p", "cmc_nonref",
                 "cmc_p", "zeg_p")


def gene_cost(N, M, d):
    """Predicted cost of one gene: the algorithmic flops of the sufficient-statistics pass."""
    return float(N) * M * (M + d + 1)


def partition_genes(Ms, world_size, N=1, d=1):
    """Static greedy longest-processing-time partition.  Returns a list of index arrays (one per rank); within a
    rank genes keep ascending order so per-rank output is already gene-ordered."""
    Ms = np.asarray(Ms, dtype=np.int64)
    order = np.argsort(-Ms, kind="stable")
    load = np.zeros(world_size)
    bins = [[] for _ in range(world_size)]
    for g in order:
        r = int(np.argmin(load))
        bins[r].append(int(g))
        load[r] += gene_cost(N, int(Ms[g]), d)
    return [np.array(sorted(b), dtype=np.int64) for b in bins]


def records_from_results(results):
    """rvt_gene_result list -> float64 array [n, len(RECORD_FIELDS)] (POD records for the gather)."""
    out = np.zeros((len(results), len(RECORD_FIELDS)), dtype=np.float64)
    for i, r in enumerate(results):
        for j, f in enumerate(RECORD_FIELDS):
            out[i, j] = float(getattr(r, f))
    return out


def broadcast_null(dist, tensors, src=0):
    """C1: broadcast the null-model tensors (X, res, v, sigma2) from `src` in place."""
    for t in tensors:
        dist.broadcast(t, src)


def gather_records(dist, local_records, counts, device=None, dst=0):
    """C2: gather the per-rank record blocks on `dst` and return them sorted by gene_id (column 0); other ranks
    get None.  `counts[r]` = number of genes of rank r (known from the static partition, so no size exchange)."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    width = local_records.shape[1]
    cap = int(max(counts)) if len(counts) else 0
    buf = torch.zeros((cap, width), dtype=torch.float64, device=device)
    if local_records.shape[0]:
        buf[: local_records.shape[0]] = torch.as_tensor(local_records, dtype=torch.float64, device=device)
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gathered, dst=dst)
    if rank != dst:
        return None
    rows = [gathered[r][: counts[r]].cpu().numpy() for r in range(world)]
    allr = np.concatenate(rows, axis=0) if rows else np.zeros((0, width))
    return allr[np.argsort(allr[:, 0], kind="stable")]
