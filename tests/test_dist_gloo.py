"""CPU, world_size = 2, gloo: the multi-GPU plumbing (static gene partition, null-model broadcast C1, ordered
gather of result records C2) without any GPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rvtests_amd import shard  # noqa: E402


def test_partition_is_balanced_and_complete():
    rng = np.random.default_rng(0)
    Ms = rng.integers(20, 81, size=2000)
    for world in (1, 2, 4, 8):
        parts = shard.partition_genes(Ms, world, N=500000, d=3)
        allg = np.sort(np.concatenate(parts))
        assert np.array_equal(allg, np.arange(len(Ms)))
        loads = [sum(shard.gene_cost(500000, int(Ms[g]), 3) for g in p) for p in parts]
        assert max(loads) / (sum(loads) / world) < 1.01
        for p in parts:
            assert np.all(np.diff(p) > 0)


class FakeResult:
    def __init__(self, gid):
        for f in shard.RECORD_FIELDS:
            setattr(self, f, 0.0)
        self.gene_id = gid
        self.skat_p = 1.0 / (1 + gid)
        self.skato_p = 2.0 / (2 + gid)


def _worker(rank, world, port, Ms, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # C1: null model broadcast
    N, d = 1000, 3
    X = torch.arange(N * d, dtype=torch.float64).reshape(N, d) if rank == 0 else torch.zeros((N, d), dtype=torch.float64)
    res = torch.linspace(0, 1, N, dtype=torch.float64) if rank == 0 else torch.zeros(N, dtype=torch.float64)
    shard.broadcast_null(dist, [X, res], src=0)
    assert float(X[-1, -1]) == N * d - 1 and abs(float(res[-1]) - 1.0) < 1e-15
    # each rank "processes" its shard, then C2
    parts = shard.partition_genes(Ms, world, N=N, d=d)
    mine = [FakeResult(int(g)) for g in parts[rank]]
    rec = shard.records_from_results(mine)
    allr = shard.gather_records(dist, rec, [len(p) for p in parts], dst=0)
    if rank == 0:
        q.put(allr)
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_ordered_gather_world2():
    Ms = list(np.random.default_rng(1).integers(5, 60, size=37))
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, Ms, q)) for r in range(2)]
    for p in procs:
        p.start()
    allr = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert allr.shape == (37, len(shard.RECORD_FIELDS))
    assert np.array_equal(allr[:, 0], np.arange(37))
    j = shard.RECORD_FIELDS.index("skat_p")
    assert np.allclose(allr[:, j], 1.0 / (1 + np.arange(37)))
