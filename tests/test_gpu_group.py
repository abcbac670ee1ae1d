"""GPU: device groups (rvt_group_*, several GPUs behind one caller) and the one-process-per-GPU bench path, exercised
on ONE device: a group that lists device 0 twice deals the gene stream to two engine contexts and must hand back the
records in submission order with the single-context values; two bench ranks share GPU 0 over gloo."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_group_orders_records_and_matches_single_context(engine):
    import rvtests_amd
    N, d = 3001, 3
    rng = np.random.default_rng(3)
    genes = []
    for g in range(75):                       # several runs per member + a ragged tail
        M = int(rng.integers(1, 70))
        Graw, G, af = synth.make_gene(N, M, seed=7000 + g, missing=0.01 if g % 4 == 0 else 0.0, common=(g % 5 == 1))
        genes.append((G, af))
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=9, G_effect=0.4 * genes[3][0][:, :2].sum(1))
    grp = rvtests_amd.Group([0, 0])
    try:
        beta, sig = grp.fit_null(0, X, y)
        ids = [1000 + 3 * g for g in range(len(genes))]
        got = []
        for g, (G, af) in enumerate(genes):
            grp.submit_gene(ids[g], G, af)
            if g == 40:                       # a partial collect in the middle of the stream
                got += grp.collect(cap=25)
        got += grp.collect()
        assert [r.gene_id for r in got] == ids
    finally:
        grp.close()
    engine.fit_null(0, X, y)
    for g, (G, af) in enumerate(genes):
        engine.submit_gene(ids[g], G, af)
    ref = engine.collect()
    assert [r.gene_id for r in ref] == ids
    for a, b in zip(got, ref):
        for f in ("skat_Q", "skat_p", "skato_Q", "skato_p", "cmc_p", "zeg_p", "cmc_nonref", "n_poly", "status"):
            assert getattr(a, f) == getattr(b, f), f
    rc, a0 = orc.skat(genes[3][0], genes[3][1], X, res, v, 0)
    assert abs(got[3].skat_p - a0.pvalue) <= 1e-6 * a0.pvalue + 1e-14


def test_group_famskat_shares(engine):
    """Related samples: the kinship is replicated on both members, host blocks are dealt in two shares that run at the
    same time, records come back in the caller's order and equal the single-context run."""
    import rvtests_amd
    from test_fam_cpu import make_family_case
    N, K, U, S, X, y = make_family_case(50, 2, 31)
    genes = [synth.make_gene(N, M, seed=400 + M, missing=0.0, common=(M % 2 == 0))[1] for M in (12, 5, 30, 7, 22, 9, 3)]
    grp = rvtests_amd.Group([0, 0])
    try:
        grp.set_kinship(U, S)
        nul = grp.fit_fam_null(X, y)
        got = grp.run_fam_tests_host(genes, tests=16, ids=list(range(50, 57)))
    finally:
        grp.close()
    eng = rvtests_amd.Engine(0)
    try:
        eng.set_kinship(U, S)
        nul1 = eng.fit_fam_null(X, y)
        assert nul1.delta == nul.delta
        ptrs = [eng.upload_block(G) for G in genes]
        ref = eng.run_fam_blocks(ptrs, [G.shape[1] for G in genes], ids=list(range(50, 57)))
    finally:
        eng.close()
    assert [r.gene_id for r in got] == list(range(50, 57))
    for a, b in zip(got, ref):
        assert a.famskat_ok == b.famskat_ok and a.famskat_Q == b.famskat_Q and a.famskat_p == b.famskat_p


def test_bench_two_ranks_share_one_gpu():
    """bench.py's N > 1 path (one process per GPU, broadcast of the null inputs, per-step gather of REAL engine records
    on rank 0) with two ranks on GPU 0 over gloo."""
    env = dict(os.environ, RVT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--samples",
           "20000", "--genes", "48", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["genes_ok"] == 48 and line["value"] > 0
    assert line["gathered_records_last_step"] == 96 and line["gathered_ids_in_order"] is True


def test_group_bgen_stream_matches_single_context(engine):
    """BGEN probability blocks through a two-member group: submission order kept, records equal the single context's."""
    import bgengen
    import rvtests_amd
    N, d = 2500, 2
    rng = np.random.default_rng(17)
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=4)
    genes = [[bgengen.layout2_block_fast(rng, N, bits=(8, 16, 32)[g % 3], missing=0.01)
              for _ in range(int(rng.integers(2, 30)))] for g in range(40)]
    grp = rvtests_amd.Group([0, 0])
    try:
        grp.fit_null(0, X, y)
        for g, blocks in enumerate(genes):
            grp.submit_gene_bgen(50 + g, blocks, 2)
        got = grp.collect()
    finally:
        grp.close()
    engine.fit_null(0, X, y)
    for g, blocks in enumerate(genes):
        engine.submit_gene_bgen(50 + g, blocks, 2, want_af=False)
    ref = engine.collect()
    assert [r.gene_id for r in got] == [r.gene_id for r in ref] == [50 + g for g in range(40)]
    for a, b in zip(got, ref):
        for f in ("skat_Q", "skat_p", "skato_p", "cmc_p", "zeg_p", "n_poly", "status"):
            assert getattr(a, f) == getattr(b, f), f


def test_group_with_submit_workers_gives_the_same_records(engine, monkeypatch):
    """RVT_GROUP_ASYNC=1: every member gets a submit worker thread (the caller only copies the buffer); records and their
    order are those of the synchronous group."""
    import rvtests_amd
    N, d = 3001, 2
    rng = np.random.default_rng(5)
    genes = []
    for g in range(70):
        M = int(rng.integers(1, 60))
        Graw, G, af = synth.make_gene(N, M, seed=9000 + g, missing=0.01 if g % 3 == 0 else 0.0, common=(g % 5 == 1))
        genes.append((Graw, G, af))
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=19)
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("RVT_GROUP_ASYNC", mode)
        grp = rvtests_amd.Group([0, 0])
        try:
            grp.fit_null(0, X, y)
            got = []
            for g, (Graw, G, af) in enumerate(genes):
                if g % 2:
                    grp.submit_gene(g, G, af)
                else:
                    grp.submit_gene_i8(g, np.where(np.isnan(Graw) | (Graw < 0), -9, Graw).astype(np.int8))
                if g == 33:
                    got += grp.collect_ready()
            got += grp.collect()
        finally:
            grp.close()
        assert [r.gene_id for r in got] == list(range(len(genes)))
        outs.append(got)
    for a, b in zip(*outs):
        for f in ("skat_Q", "skat_p", "skato_p", "cmc_p", "zeg_p", "cmc_nonref", "n_poly", "status"):
            assert getattr(a, f) == getattr(b, f), f


def test_group_meta_score_and_cov_band_equal_the_single_context(engine):
    """`--meta score` / `--meta cov` over a 2-member group (both on device 0): shares / chunks with a one-window halo,
    no exchange — the numbers of the single-context calls on the whole block."""
    import rvtests_amd
    N, V, d, halo = 3001, 700, 3, 150
    rng = np.random.default_rng(12)
    G = np.asfortranarray(rng.binomial(2, 10 ** rng.uniform(-2.5, -0.6, V), size=(N, V)).astype(np.float64))
    G[:, 5] = 0.0                                           # a monomorphic site
    G[rng.random(N) < 0.01, 9] = G[:, 9].mean()             # an imputed column
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=9)
    engine.fit_null(0, X, y)
    ptr = engine.upload_block(G)
    sc = engine.score_block(ptr, V)
    ok0, u0, v0, e0, se0, p0 = sc["ok"], sc["U"], sc["V"], sc["effect"], sc["se"], sc["p"]
    cov0, xz0, zz0, poly0 = engine.cov_block(ptr, V)
    engine.free_block(ptr)
    grp = rvtests_amd.Group([0, 0])
    try:
        grp.fit_null(0, X, y)
        ok, u, vv, e, se, p = grp.score_block_host(G)
        band, xz, zz, poly = grp.cov_band_host(G, d, halo, chunk=200)
    finally:
        grp.close()
    assert (ok == ok0).all() and (poly == poly0).all()
    good = ok0 == 1
    for a, b in ((u, u0), (vv, v0), (e, e0), (se, se0), (p, p0)):
        assert np.allclose(a[good], b[good], rtol=1e-11, atol=0)
    scale = np.nanmax(np.abs(cov0[np.triu_indices(V)]))
    for h in range(V):
        for t in range(halo + 1):
            j = h + t
            if j >= V:
                assert np.isnan(band[h, t])
            else:
                assert abs(band[h, t] - cov0[h, j]) <= 1e-9 * scale, (h, j)
    assert np.allclose(xz, xz0, rtol=1e-9, atol=1e-9 * max(np.abs(xz0).max(), 1.0)) and np.allclose(zz, zz0, rtol=1e-12)


def test_group_on_two_devices_matches_single_context(engine):
    """Members on DIFFERENT devices (the case the group exists for): feeder threads are on by default, the 2-bit and int8
    hand-offs are dealt over both GPUs, records come back in submission order with the single-context values, and SKAT
    permutations take the counter-based mode (genes dealt over members cannot share one rand() stream).  Skips on a box with
    one GPU — every box this suite has run on so far; it runs the day a multi-GPU lease exists."""
    import torch
    import rvtests_amd
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    N, d = 4003, 2
    rng = np.random.default_rng(12)
    genes = []
    for g in range(96):
        M = int(rng.integers(1, 70))
        raw = rng.binomial(2, 10 ** rng.uniform(-2.5, -0.7, M), size=(N, M)).astype(np.int8)
        if g % 3 == 0:
            raw[rng.random((N, M)) < 0.01] = -9
        genes.append(np.asfortranarray(raw))
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=5)
    grp = rvtests_amd.Group([0, 1])
    try:
        grp.fit_null(0, X, y)
        for g, raw in enumerate(genes):
            if g % 2:
                grp.submit_gene_bed(g, rvtests_amd.Engine.pack_bed(raw.astype(np.float64)), raw.shape[1])
            else:
                grp.submit_gene_i8(g, raw)
        got = grp.collect()
    finally:
        grp.close()
    assert [r.gene_id for r in got] == list(range(len(genes)))
    engine.fit_null(0, X, y)
    for g, raw in enumerate(genes):
        engine.submit_gene_raw(g, raw, want_af=False)
    ref = engine.collect()
    for a, b in zip(got, ref):
        for f in ("skat_Q", "skat_p", "skato_p", "cmc_p", "zeg_p", "cmc_nonref", "n_poly", "status"):
            assert getattr(a, f) == getattr(b, f), f


def test_bench_eight_ranks_share_one_gpu():
    """bench.py --gpus 8 as the driver launches it on an 8-GPU node, with all eight ranks on GPU 0 over gloo (the one device a
    1-GPU lease has): eight ranks seen, every rank's gene shard computed (genes_ok counts rank 0's), the per-step C2 gather of
    real engine records complete (8 x genes) and in gene order, a rate and a gather time per rank.  What is left for the first
    real 8-GPU run to show is RCCL over xGMI itself, not the plumbing."""
    env = dict(os.environ, RVT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--samples",
           "12000", "--genes", "24", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["config"]["genes_ok"] == 24 and line["value"] > 0
    assert line["gathered_records_last_step"] == 8 * 24 and line["gathered_ids_in_order"] is True
    assert len(line["per_rank_gene_sets_per_s"]) == 8 and min(line["per_rank_gene_sets_per_s"]) > 0
    assert len(line["c2_gather_ms_per_step_by_rank"]) == 8
    assert line["scaling"] == "weak" and abs(line["value"] - 8 * 24 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]


def test_group_of_eight_members_equals_the_single_context(engine):
    """rvt_group with EIGHT members (the one-thread C++ caller's form of the 8-GPU node; all on device 0 here): the gene stream
    dealt in runs over eight contexts comes back in submission order with the single context's records, and `--meta score` /
    `--meta cov` over eight shares with a one-window halo give the single context's numbers."""
    import rvtests_amd
    N, d = 2501, 2
    rng = np.random.default_rng(8)
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=6)
    genes = [synth.make_gene(N, int(rng.integers(1, 40)), seed=900 + g, missing=0.01, common=True, mono=True)[1:] for g in range(300)]
    grp = rvtests_amd.Group([0] * 8)
    try:
        grp.fit_null(0, X, y)
        for g, (G, af) in enumerate(genes):
            grp.submit_gene(1000 + g, G, af)
        got = grp.collect()
        V, halo = 640, 90
        Gm = np.asfortranarray(rng.binomial(2, 10 ** rng.uniform(-2.5, -0.6, V), size=(N, V)).astype(np.float64))
        Gm[:, 3] = 2.0
        sc = grp.score_block_host(Gm)
        band, xz, zz, poly = grp.cov_band_host(Gm, d, halo, chunk=70)
    finally:
        grp.close()
    engine.fit_null(0, X, y)
    for g, (G, af) in enumerate(genes):
        engine.submit_gene(1000 + g, G, af)
    ref = engine.collect()
    assert [r.gene_id for r in got] == [r.gene_id for r in ref] == [1000 + g for g in range(300)]
    for a, b in zip(got, ref):
        for f in ("skat_Q", "skat_p", "skato_p", "cmc_p", "zeg_p", "n_poly", "status", "cmc_nonref"):
            assert getattr(a, f) == getattr(b, f), (a.gene_id, f)
    ptr = engine.upload_block(Gm)
    s0 = engine.score_block(ptr, V)
    cov0, xz0, zz0, poly0 = engine.cov_block(ptr, V)
    ok, u, vv, e, se, p_ = sc
    assert (ok == s0["ok"]).all() and (poly == poly0).all() and not poly[3]
    good = s0["ok"] == 1
    for a, b in ((u, s0["U"]), (vv, s0["V"]), (e, s0["effect"]), (se, s0["se"]), (p_, s0["p"])):
        assert np.allclose(a[good], b[good], rtol=1e-11, atol=0)
    scale = np.nanmax(np.abs(cov0[np.triu_indices(V)]))
    for h in range(0, V, 3):
        w = min(halo + 1, V - h)
        assert np.abs(band[h, :w] - cov0[h, h:h + w]).max() <= 1e-9 * scale
        assert np.isnan(band[h, w:]).all()
