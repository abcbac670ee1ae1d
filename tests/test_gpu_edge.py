"""Edge cases of the hot path through the C ABI: many covariates (tile shapes without an unrolled body go to the
panelled kernel), tiny sample counts, empty batches, single-variant and all-monomorphic genes, oversize genes."""
import numpy as np
import pytest

import orc
import synth
from test_gpu_parity import _check_gene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("binary,d", [(0, 16), (0, 9), (1, 1), (0, 13)])
def test_many_covariates_and_all_tile_shapes(engine, binary, d):
    """d up to RVT_MAX_COV: M + d + 1 can need two column tiles more than row tiles — (1,3), (2,4), ... have no
    unrolled body and must take the panelled kernel, in one batch with the ordinary classes."""
    N = 2500
    Ms = (1, 3, 15, 16, 17, 31, 32, 40, 63, 64, 80, 95, 96, 100)
    genes = [synth.make_gene(N, M, seed=7 * M + d, missing=0.01 if M % 2 else 0.0, common=(M % 3 == 0),
                             mono=(M > 4), maf_hi=-1.0)[1:] for M in Ms]
    X, y, res, v, s2 = synth.make_null(N, d, binary, seed=3 + d, G_effect=0.4 * genes[5][0][:, :3].sum(1))
    engine.set_null(binary, X, res, v, s2)
    ptrs = [engine.upload_block(G) for G, af in genes]
    out = engine.run_blocks(ptrs, [G.shape[1] for G, af in genes], [af for G, af in genes])
    for p in ptrs:
        engine.free_block(p)
    for r, (G, af) in zip(out, genes):
        _check_gene(r, G, af, X, y, res, v, binary, d)


@pytest.mark.parametrize("N", [15, 17, 33])
def test_tiny_sample_counts(engine, N):
    """fewer samples than one 16-sample step / not a multiple of 16 (guarded last step, rank <= N eigenvalue rule)"""
    rng = np.random.default_rng(N)
    genes = []
    for M in (2, 5, 20):
        G = rng.integers(0, 3, size=(N, M)).astype(np.float64)
        G[:, 0] = rng.permutation(np.r_[np.ones(3), np.zeros(N - 3)])
        af = 0.5 * G.sum(0) / N
        genes.append((G, af))
    X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=N)
    engine.set_null(0, X, res, v, s2)
    ptrs = [engine.upload_block(G) for G, af in genes]
    out = engine.run_blocks(ptrs, [G.shape[1] for G, af in genes], [af for G, af in genes])
    for r, (G, af) in zip(out, genes):
        _check_gene(r, G, af, X, y, res, v, 0, 2)


def test_empty_batch_and_degenerate_genes(engine):
    import rvtests_amd
    N = 400
    X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=1)
    engine.set_null(0, X, res, v, s2)
    # n = 0 is a no-op
    assert engine.L.rvt_run_blocks(engine.ctx, 0, None, None, None, None, 15, None, None) == 0
    genes = [(np.zeros((N, 4)), np.zeros(4)),                       # all monomorphic (zero)
             (np.full((N, 2), 2.0), np.ones(2)),                    # all monomorphic (non-zero)
             (np.r_[np.ones(5), np.zeros(N - 5)].reshape(N, 1), np.array([2.5 / N]))]   # single variant
    ptrs = [engine.upload_block(G) for G, af in genes]
    out = engine.run_blocks(ptrs, [G.shape[1] for G, af in genes], [af for G, af in genes])
    for r, (G, af) in zip(out, genes):
        _check_gene(r, G, af, X, y, res, v, 0, 2)
    assert out[0].n_poly == 0 and out[1].n_poly == 0 and out[2].n_poly == 1


def test_oversize_gene_is_refused(engine):
    import rvtests_amd
    N = 64
    X, y, res, v, s2 = synth.make_null(N, 1, 0, seed=2)
    engine.set_null(0, X, res, v, s2)
    M = 1025                                                         # RVT_MAX_VARIANTS + 1
    G = np.random.default_rng(0).integers(0, 3, size=(N, M)).astype(np.float64)
    ptr = engine.upload_block(G)
    with pytest.raises(rvtests_amd.RvtError):
        engine.run_blocks([ptr], [M], [0.5 * G.sum(0) / N])


def test_large_gene_list_is_chunked(engine):
    """rvt_run_blocks splits long lists into batches of 256 genes; records come back in the caller's order."""
    N, n = 300, 600
    X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=5)
    engine.set_null(0, X, res, v, s2)
    genes = [synth.make_gene(N, 1 + (g % 7), seed=g, missing=0.0, common=True, mono=False)[1:] for g in range(n)]
    ptrs = [engine.upload_block(G) for G, af in genes]
    out = engine.run_blocks(ptrs, [G.shape[1] for G, af in genes], [af for G, af in genes])
    assert [r.gene_id for r in out] == list(range(n))
    for g in (0, 255, 256, 257, 511, 512, 599):
        r, (G, af) = out[g], genes[g]
        rc, a = orc.skat(G, af, X, res, v, 0)
        assert r.n_variants == G.shape[1] and r.n_poly == a.n_poly
        if a.n_poly:
            assert abs(r.skat_Q - a.Q) <= 1e-10 * a.Q


@pytest.mark.parametrize("want_af", [True, False])
@pytest.mark.parametrize("packed", [False, True, "bed"])
def test_raw_and_packed_submission_consolidate_on_device(engine, packed, want_af):
    """rvt_submit_gene_raw / _i8 / _bed: counter allele frequencies and mean imputation done on the device give the same
    records as handing over the block DataConsolidator would have produced (oracle imputation + counter AF).  N is not
    a multiple of 4 so the last byte of every PLINK row is partly used."""
    N = 1202
    X, y, res, v, s2 = synth.make_null(N, 3, 0, seed=9)
    engine.set_null(0, X, res, v, s2)
    cases = [synth.make_gene(N, M, seed=70 + M, missing=miss, common=True, mono=True)
             for M, miss in ((5, 0.0), (18, 0.03), (40, 0.2), (3, 0.9))]
    rng = np.random.default_rng(4)
    if not packed:      # fractional dosages with missing values: the reference's truncating running allele count
        Graw = np.round(rng.uniform(0, 2, size=(N, 6)), 3)
        Graw[rng.random((N, 6)) < 0.05] = -9.0
        Graw[:, 2] = np.where(Graw[:, 2] > 1.9, 2.5, Graw[:, 2])       # > 2: counted missing, not imputed
        cases.append((Graw, orc.impute_mean(Graw), orc.counter_af(Graw)))
    for k, (Graw, G, af) in enumerate(cases):
        # want_af = False: the call returns without waiting for the device; the frequencies are picked up when the
        # gene's group is launched
        if packed == "bed":
            got_af = engine.submit_gene_bed(k, engine.pack_bed(Graw), Graw.shape[1], want_af=want_af)
        else:
            raw = Graw.astype(np.int8) if packed else Graw
            got_af = engine.submit_gene_raw(k, raw, want_af=want_af)
        if want_af:
            assert np.allclose(got_af, af, rtol=1e-14, atol=0)
    got = engine.collect()
    ptrs = [engine.upload_block(G) for Graw, G, af in cases]
    want = engine.run_blocks(ptrs, [G.shape[1] for Graw, G, af in cases], [af for Graw, G, af in cases])
    for a, b in zip(got, want):
        for f in ("n_variants", "n_poly", "skat_ok", "skato_ok", "cmc_ok", "cmc_nonref", "zeg_ok", "skato_rho"):
            assert getattr(a, f) == getattr(b, f), f
        for f in ("skat_Q", "skat_p", "skato_Q", "skato_p", "cmc_p", "zeg_p"):
            x, w = getattr(a, f), getattr(b, f)
            # identical blocks; the allele frequency of a fractional-dosage column may differ in its last bit (parallel
            # instead of sequential sum), hence not exact equality
            assert abs(x - w) <= 1e-11 * abs(w) + 1e-300, f
