"""GPU: long streams through every streaming entry point.  More genes than one launch group (16) are in flight, so blocks are
decoded / consolidated on the transfer stream WHILE earlier groups compute — the records must be those of the same genes
submitted as finished fp64 blocks, whatever the entry point."""
import zlib

import numpy as np
import pytest

import bgengen
import orc
import synth
import vcfgen

pytestmark = pytest.mark.gpu
FIELDS = ("status", "n_poly", "skat_Q", "skat_p", "skato_p", "cmc_p", "zeg_p", "cmc_nonref")


@pytest.fixture
def eng():
    import rvtests_amd
    e = rvtests_amd.Engine(0)
    yield e
    e.close()


def _reference(eng, mats):
    """Records of the raw matrices (missing = -9) through the host-side consolidation of the oracle + rvt_submit_gene."""
    for g, raw in enumerate(mats):
        G = orc.impute_mean(raw)
        eng.submit_gene(g, G, orc.counter_af(raw))
    return eng.collect()


@pytest.mark.parametrize("mode", ["raw", "i8", "bed", "vcf", "bgen"])
def test_long_stream_equals_block_submission(eng, mode):
    rng = np.random.default_rng(zlib.crc32(mode.encode()) % 1000)   # (str hashes change from process to process)
    N, d, n_genes = 3000, 2, 45
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=6)
    eng.set_null(0, X, res, v, s2)
    eng.vcf_set_samples(np.arange(N, dtype=np.int32))
    mats, payload = [], []
    for g in range(n_genes):
        M = int(rng.integers(1, 40))
        if mode == "bgen":
            blocks = [bgengen.layout2_block_fast(rng, N, bits=(8, 16)[g % 2], missing=0.02) for _ in range(M)]
            mats.append(np.asfortranarray(np.column_stack([orc.bgen_block_genotypes(b, 2, N) for b in blocks])))
            payload.append(blocks)
        else:
            maf = 10 ** rng.uniform(-2.5, -0.7, M)
            raw = rng.binomial(2, maf, size=(N, M)).astype(np.float64)
            raw[rng.random((N, M)) < 0.01] = -9.0
            mats.append(np.asfortranarray(raw))
            if mode == "vcf":
                payload.append([vcfgen.fixed_width_record(raw[:, j].astype(np.int64), pos=100 + j) for j in range(M)])
            elif mode == "bed":
                payload.append(eng.pack_bed(raw))
            else:
                payload.append(raw)
    for g in range(n_genes):
        if mode == "raw":
            eng.submit_gene_raw(g, payload[g], want_af=False)
        elif mode == "i8":
            eng.submit_gene_raw(g, payload[g].astype(np.int8), want_af=False)
        elif mode == "bed":
            eng.submit_gene_bed(g, payload[g], mats[g].shape[1], want_af=False)
        elif mode == "vcf":
            eng.submit_gene_vcf(g, payload[g], want_af=False)
        else:
            eng.submit_gene_bgen(g, payload[g], 2, want_af=False)
    got = eng.collect()
    ref = _reference(eng, mats)
    assert [r.gene_id for r in got] == list(range(n_genes)) == [r.gene_id for r in ref]
    assert sum(r.n_poly for r in got) > 0
    for a, b in zip(got, ref):
        for f in FIELDS:
            x, y_ = getattr(a, f), getattr(b, f)
            assert x == y_ or abs(x - y_) <= 1e-12 * abs(y_), (mode, a.gene_id, f, x, y_)


def test_mixed_entry_points_with_partial_collects(eng):
    """One stream that mixes every entry point gene by gene, with rvt_collect_ready calls in between and blocks going
    back to the pool and out again: order kept, records those of the finished-block submission."""
    rng = np.random.default_rng(77)
    N, d, n_genes = 2011, 3, 130                       # odd N: pad rows in every block
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=8)
    eng.set_null(0, X, res, v, s2)
    eng.vcf_set_samples(np.arange(N, dtype=np.int32))
    mats, got = [], []
    for g in range(n_genes):
        M = int(rng.integers(1, 60))
        mode = ("raw", "i8", "bed", "vcf", "bgen", "block")[int(rng.integers(0, 6))]
        if mode == "bgen":
            blocks = [bgengen.layout2_block_fast(rng, N, bits=16, missing=0.02) for _ in range(M)]
            raw = np.asfortranarray(np.column_stack([orc.bgen_block_genotypes(b, 2, N) for b in blocks]))
            eng.submit_gene_bgen(g, blocks, 2, want_af=False)
        else:
            maf = 10 ** rng.uniform(-2.5, -0.5, M)
            raw = np.asfortranarray(rng.binomial(2, maf, size=(N, M)).astype(np.float64))
            raw[rng.random((N, M)) < 0.01] = -9.0
            if mode == "raw":
                eng.submit_gene_raw(g, raw, want_af=bool(g % 2))
            elif mode == "i8":
                eng.submit_gene_raw(g, raw.astype(np.int8), want_af=False)
            elif mode == "bed":
                eng.submit_gene_bed(g, eng.pack_bed(raw), M, want_af=False)
            elif mode == "vcf":
                eng.submit_gene_vcf(g, [vcfgen.fixed_width_record(raw[:, j].astype(np.int64), pos=9 + j) for j in range(M)],
                                    want_af=False)
            else:
                eng.submit_gene(g, orc.impute_mean(raw), orc.counter_af(raw))
        mats.append(raw)
        if g % 23 == 22:
            got += eng.collect_ready()
    got += eng.collect()
    ref = _reference(eng, mats)
    assert [r.gene_id for r in got] == list(range(n_genes))
    for a, b in zip(got, ref):
        for f in FIELDS:
            x, y_ = getattr(a, f), getattr(b, f)
            assert x == y_ or abs(x - y_) <= 1e-12 * abs(y_), (a.gene_id, f, x, y_)


def test_long_stream_binary_trait(eng):
    """The same under a logistic null model: streamed hard-call blocks take the weighted int8 kernel, the genes with
    imputed means included (round 4: suffstat_hcx.hip.h keeps them; the one-wave kernel handed them back to the fp64
    kernel); records equal those of the finished-block submission."""
    rng = np.random.default_rng(5)
    N, d, n_genes = 2500, 2, 40
    X, y, res, v, s2 = synth.make_null(N, d, 1, seed=10)
    eng.set_null(1, X, res, v, s2)
    mats = []
    eng.set_profiling(True)
    eng.timing(reset=True)
    for g in range(n_genes):
        M = int(rng.integers(1, 70))
        maf = 10 ** rng.uniform(-2.5, -0.7, M)
        raw = np.asfortranarray(rng.binomial(2, maf, size=(N, M)).astype(np.float64))
        if g % 3 == 0:
            raw[rng.random((N, M)) < 0.01] = -9.0          # imputed means: handed back to the fp64 kernel
        mats.append(raw)
        eng.submit_gene_raw(g, raw.astype(np.int8), want_af=False)
    got = eng.collect()
    tm = eng.timing(reset=True)
    eng.set_profiling(False)
    assert tm.genes_hard_call == n_genes and tm.genes_handed_back == 0
    ref = _reference(eng, mats)
    for a, b in zip(got, ref):
        for f in FIELDS:
            x, y_ = getattr(a, f), getattr(b, f)
            assert x == y_ or abs(x - y_) <= 1e-9 * abs(y_), (a.gene_id, f, x, y_)


def test_copy_pool_scales_on_the_gpu_host():
    """The staged copies (host_stage.h) are filled by a pool of threads: on the GPU box's host several threads must move
    clearly more than one does (one core copies ~10 GB/s, the link takes ~55)."""
    import ctypes as C
    import hc
    L = hc.lib()
    L.hc_copy_rate.restype = C.c_double
    L.hc_copy_rate.argtypes = [C.c_size_t, C.c_int, C.c_int]
    r1 = L.hc_copy_rate(256 << 20, 1, 4)
    r8 = L.hc_copy_rate(256 << 20, 8, 4)
    print("copy pool on this host: 1 thread %.1f GB/s, 8 threads %.1f GB/s" % (r1, r8))
    assert r8 > 2.0 * r1


def test_staged_and_plain_copies_give_the_same_records(monkeypatch):
    """RVT_STAGE=0 (the runtime's own pageable copies) against the staged copies: identical records, every entry point."""
    import rvtests_amd
    rng = np.random.default_rng(8)
    N, d = 40_000, 2
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=3)
    genes = []
    for g in range(20):
        M = int(rng.integers(3, 60))
        raw = np.asfortranarray(rng.binomial(2, 10 ** rng.uniform(-2.5, -0.7, M), size=(N, M)).astype(np.float64))
        if g % 3 == 0:
            raw[rng.random((N, M)) < 0.01] = -9.0
        genes.append(raw)
    outs = []
    for stage in ("1", "0"):
        monkeypatch.setenv("RVT_STAGE", stage)
        e = rvtests_amd.Engine(0)
        e.fit_null(0, X, y)
        for g, raw in enumerate(genes):
            if g % 4 == 0:
                e.submit_gene(g, orc.impute_mean(raw), orc.counter_af(raw))
            elif g % 4 == 1:
                e.submit_gene_raw(g, raw, want_af=False)
            elif g % 4 == 2:
                e.submit_gene_raw(g, raw.astype(np.int8), want_af=(g % 8 == 2))
            else:
                e.submit_gene_bed(g, e.pack_bed(raw), raw.shape[1], want_af=False)
        outs.append(e.collect())
        e.close()
    for a, b in zip(*outs):
        for f in FIELDS:
            assert getattr(a, f) == getattr(b, f), (a.gene_id, f)


def test_registered_hand_off_buffer_is_read_before_the_call_returns():
    """rvt_host_register: ONE page-locked buffer, refilled with the next gene as soon as the submission returns (the
    reference's loop, src/Main.cpp:1086,1225) — fp64, raw, int8 and 2-bit hand-offs; identical records to the same stream
    from ordinary pageable arrays; a group registers the range once for both members."""
    import rvtests_amd
    rng = np.random.default_rng(18)
    N, d = 60_000, 2
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=3)
    genes = []
    for g in range(24):
        M = int(rng.integers(3, 60))
        raw = np.asfortranarray(rng.binomial(2, 10 ** rng.uniform(-2.5, -0.7, M), size=(N, M)).astype(np.float64))
        if g % 3 == 0:
            raw[rng.random((N, M)) < 0.01] = -9.0
        genes.append(raw)
    buf = np.zeros(N * 60 * 8 + 4096, dtype=np.uint8)            # the caller's reused buffer (bytes)

    def view(dtype, shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return buf[:n].view(dtype).reshape(shape, order="F")

    def stream(e, registered):
        for g, raw in enumerate(genes):
            M = raw.shape[1]
            if g % 4 == 0:
                src = orc.impute_mean(raw)
                dst = view(np.float64, (N, M)) if registered else src
                dst[...] = src
                e.submit_gene(g, dst, orc.counter_af(raw))
            elif g % 4 == 1:
                dst = view(np.float64, (N, M)) if registered else raw
                dst[...] = raw
                e.submit_gene_raw(g, dst, want_af=False)
            elif g % 4 == 2:
                r8 = raw.astype(np.int8)
                dst = view(np.int8, (N, M)) if registered else r8
                dst[...] = r8
                e.submit_gene_raw(g, dst, want_af=(g % 8 == 2))
            else:
                bed = e.pack_bed(raw)
                dst = buf[:bed.size].reshape(bed.shape) if registered else bed
                dst[...] = bed
                e.submit_gene_bed(g, dst, M, want_af=False)
            if registered:
                buf[:N * M * 8] = 0xA5                            # the caller moves on at once
        return e.collect()

    e = rvtests_amd.Engine(0)
    e.fit_null(0, X, y)
    want = stream(e, False)
    e.host_register(buf)
    with pytest.raises(rvtests_amd.RvtError):
        e.host_register(buf[100:200])                             # overlaps a registered range
    got = stream(e, True)
    e.host_unregister(buf)
    with pytest.raises(rvtests_amd.RvtError):
        e.host_unregister(buf)
    again = stream(e, False)
    e.close()
    for a, b, c_ in zip(got, want, again):
        for f in FIELDS:
            assert getattr(a, f) == getattr(b, f) == getattr(c_, f), (a.gene_id, f)
    grp = rvtests_amd.Group([0, 0])
    try:
        grp.fit_null(0, X, y)
        grp.host_register(buf)
        for g, raw in enumerate(genes[:16]):
            M = raw.shape[1]
            r8 = raw.astype(np.int8)
            dst = view(np.int8, (N, M))
            dst[...] = r8
            grp.submit_gene_i8(g, dst)
            buf[:N * M] = 0x5A
        out = grp.collect()
        grp.host_unregister(buf)
    finally:
        grp.close()
    e = rvtests_amd.Engine(0)
    e.fit_null(0, X, y)
    for g, raw in enumerate(genes[:16]):
        e.submit_gene_raw(g, raw.astype(np.int8), want_af=False)
    ref = e.collect()
    e.close()
    for a, b in zip(out, ref):
        for f in FIELDS:
            assert getattr(a, f) == getattr(b, f), (a.gene_id, f)


def test_batched_submit_equals_gene_by_gene(eng):
    """rvt_submit_genes (several genes per call: int8 matrices and PLINK 2-bit rows) gives the records of the single-gene
    entry points."""
    import rvtests_amd
    rng = np.random.default_rng(17)
    N, d = 3001, 2
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=4)
    eng.set_null(0, X, res, v, s2)
    raws = []
    for g in range(21):
        M = int(rng.integers(1, 80))
        raw = rng.binomial(2, 10 ** rng.uniform(-2.5, -0.7, M), size=(N, M)).astype(np.int8)
        if g % 2:
            raw[rng.random((N, M)) < 0.01] = -9
        raws.append(np.asfortranarray(raw))
    for g, raw in enumerate(raws):
        eng.submit_gene_raw(g, raw, want_af=False)
    ref = eng.collect()
    eng.submit_genes(2, range(len(raws)), raws, [r.shape[1] for r in raws])
    got8 = eng.collect()
    beds = [eng.pack_bed(r.astype(np.float64)) for r in raws]
    eng.submit_genes(3, range(len(raws)), beds, [r.shape[1] for r in raws])
    gotb = eng.collect()
    for a, b, c in zip(ref, got8, gotb):
        for f in ("gene_id", "skat_Q", "skat_p", "skato_p", "cmc_p", "zeg_p", "cmc_nonref", "n_poly", "status"):
            assert getattr(a, f) == getattr(b, f), f
        for f in ("gene_id", "cmc_nonref", "n_poly", "status"):
            assert getattr(a, f) == getattr(c, f), f
        for f in ("skat_Q", "skat_p", "skato_p", "cmc_p", "zeg_p"):
            x, y_ = getattr(a, f), getattr(c, f)
            assert abs(x - y_) <= 1e-9 * abs(x) + 1e-300, f      # (the packed-row kernel sums in another order)


def test_fp64_boundary_packed_on_the_way_gives_the_same_records(monkeypatch):
    """rvt_submit_gene (the block of doubles the reference's gene loop hands over) packed to 2-bit rows by the staging threads
    (host_stage.h pack_column_f64; RVT_PACK_FP64=0 switches it off): hard-call genes, genes with mean-imputed columns (one
    other value per column), a column that is entirely the other value, N not a multiple of 4 / 16 — identical records; a
    dosage gene and a gene with two other values in one column cross as doubles and give identical records as well."""
    import rvtests_amd
    rng = np.random.default_rng(18)
    N, d = 40_003, 2
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=5)
    genes = []
    for g in range(14):
        M = int(rng.integers(3, 90))
        raw = np.asfortranarray(rng.binomial(2, 10 ** rng.uniform(-2.5, -0.7, M), size=(N, M)).astype(np.float64))
        if g % 2 == 0:
            raw[rng.random((N, M)) < 0.01] = -9.0
        if g == 4:
            raw[:, 1] = -9.0                                 # every call missing: the column is one other value (0.0 here)
        G = orc.impute_mean(raw)
        af = orc.counter_af(raw)
        if g == 6:
            G[5, 2] = 0.25                                   # a second other value in an imputed column: not representable
        if g == 9:
            G = np.asfortranarray(np.round(G + rng.uniform(0, 0.3, size=G.shape) * (G > 0), 3))   # dosages
        genes.append((G, af))
    outs, packed = [], []
    for sw in ("1", "0"):
        monkeypatch.setenv("RVT_PACK_FP64", sw)
        e = rvtests_amd.Engine(0)
        e.fit_null(0, X, y)
        e.set_profiling(True)
        for g, (G, af) in enumerate(genes):
            e.submit_gene(g, G, af)
        outs.append(e.collect())
        packed.append(e.timing(reset=True).genes_hard_call)
        e.close()
    assert [r.gene_id for r in outs[0]] == list(range(len(genes)))
    for a, b in zip(*outs):
        for f in FIELDS:
            x, y_ = getattr(a, f), getattr(b, f)
            assert x == y_ or (x != x and y_ != y_), (a.gene_id, f, x, y_)
