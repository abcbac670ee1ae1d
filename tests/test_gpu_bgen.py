"""GPU: the BGEN front end (rvt_bgen_decode / rvt_submit_gene_bgen) against the oracle's restatement of the reference's
block parsers and genotype rule (oracle/orc_bgen.cpp, pinned to the reference's golden files in test_bgen_cpu.py) — the
decoded doubles bit-exact, and the tests' records identical to those of the raw hand-off fed with the oracle-decoded
matrix."""
import base64
import json
import os
import zlib

import numpy as np
import pytest

import bgengen
import orc
import synth

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bgen_blocks.json")))


@pytest.fixture
def eng():
    import rvtests_amd
    e = rvtests_amd.Engine(0)
    yield e
    e.close()


def _oracle_matrix(blocks, layout, n_file, rows=None, n_rows=None):
    n_rows = n_file if n_rows is None else n_rows
    out = np.full((n_rows, len(blocks)), -9.0, order="F")
    for j, b in enumerate(blocks):
        g = orc.bgen_block_genotypes(b, layout, n_file)
        if rows is None:
            out[:, j] = g
        else:
            out[rows[rows >= 0], j] = g[rows >= 0]
    return out


def test_reference_golden_blocks_decode_bit_exactly(eng):
    by_file = {}
    for c in GOLD["cases"]:
        by_file.setdefault((c["file"], c["layout"], c["N"]), []).append(zlib.decompress(base64.b64decode(c["block"])))
    assert len(by_file) == 5
    for (name, layout, N), blocks in by_file.items():
        got = eng.bgen_decode(blocks, layout, N)
        want = _oracle_matrix(blocks, layout, N)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), name


@pytest.mark.parametrize("bits", [1, 2, 3, 5, 7, 8, 9, 12, 16, 17, 23, 24, 25, 31, 32])
def test_every_bit_width_and_shape_bit_exactly(eng, bits):
    rng = np.random.default_rng(100 + bits)
    N = 3001
    blocks = [bgengen.layout2_block(rng, N, bits, missing=0.05),
              bgengen.layout2_block(rng, N, bits, phased=True, missing=0.05),
              bgengen.layout2_block(rng, N, bits, K=3, missing=0.02),
              bgengen.layout2_block(rng, N, bits, K=1, missing=0.1),
              bgengen.layout2_block(rng, N, bits, missing=0.02, haploid=0.3, odd=0.1),
              bgengen.layout2_block(rng, N, bits, phased=True, K=3, missing=0.02, haploid=0.2, odd=0.2),
              bgengen.layout2_block(rng, N, bits, K=4, ploidy=3, missing=0.0, haploid=0.3)]
    got = eng.bgen_decode(blocks, 2, N)
    want = _oracle_matrix(blocks, 2, N)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert (want != -9.0).any()


def test_sample_map_and_layout1(eng):
    rng = np.random.default_rng(7)
    n_file, n_keep = 6000, 4500
    rows = np.full(n_file, -1, dtype=np.int32)
    keep = rng.choice(n_file, n_keep, replace=False)
    rows[keep] = rng.permutation(n_keep)
    eng.vcf_set_samples(rows)
    blocks = [bgengen.layout1_block(rng, n_file) for _ in range(5)]
    got = eng.bgen_decode(blocks, 1, n_keep)
    want = _oracle_matrix(blocks, 1, n_file, rows, n_keep)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    blocks = [bgengen.layout2_block(rng, n_file, 11, missing=0.03, haploid=0.1) for _ in range(4)]
    got = eng.bgen_decode(blocks, 2, n_keep)
    want = _oracle_matrix(blocks, 2, n_file, rows, n_keep)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_submit_gene_bgen_equals_raw_handoff(eng):
    """The gene tests on BGEN blocks = the gene tests on the oracle-decoded dosage matrix handed over as raw doubles."""
    rng = np.random.default_rng(21)
    N, d = 4000, 3
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=5)
    eng.set_null(0, X, res, v, s2)
    genes = []
    for g in range(6):
        M = int(rng.integers(3, 40))
        layout = 1 if g % 3 == 2 else 2
        if layout == 1:
            blocks = [bgengen.layout1_block(rng, N) for _ in range(M)]
        else:
            blocks = [bgengen.layout2_block_fast(rng, N, bits=(8, 16, 32)[g % 3], missing=0.02) for _ in range(M)]
        genes.append((blocks, layout))
    afs = []
    for g, (blocks, layout) in enumerate(genes):
        afs.append(eng.submit_gene_bgen(g, blocks, layout))
    got = eng.collect()
    for g, (blocks, layout) in enumerate(genes):
        eng.submit_gene_raw(100 + g, _oracle_matrix(blocks, layout, N))
    want = eng.collect()
    assert [r.gene_id for r in got] == list(range(6))
    for a, b in zip(got, want):
        assert a.n_poly == b.n_poly and a.status == b.status
        for f in ("skat_Q", "skat_p", "skato_p"):
            assert getattr(a, f) == getattr(b, f), f
        for f in ("cmc_p", "zeg_p"):  # (handed-back raw doubles: burden sums in burden_fallback_kernel's order)
            assert abs(getattr(a, f) - getattr(b, f)) <= 1e-12 * abs(getattr(b, f)) + 1e-7 * (getattr(b, f) > 0.999), f
    # the counter frequencies (GenotypeCounter on dosages) against the oracle's consolidation
    blocks, layout = genes[0]
    raw = _oracle_matrix(blocks, layout, N)
    assert np.allclose(afs[0], orc.counter_af(raw), rtol=0, atol=1e-15)


def test_short_block_is_reported(eng):
    rng = np.random.default_rng(3)
    N = 2000
    good = bgengen.layout2_block(rng, N, 16, missing=0.0)
    with pytest.raises(Exception):
        eng.bgen_decode([good[:-40]], 2, N)
    with pytest.raises(Exception):
        eng.bgen_decode([good], 2, N + 1)          # sample count of the block differs
    lied = bytearray(good)
    lied[7] = 1                                    # declares haploid at most, holds diploid samples
    with pytest.raises(Exception):
        eng.bgen_decode([bytes(lied)], 2, N)
    assert np.array_equal(eng.bgen_decode([good], 2, N), _oracle_matrix([good], 2, N))


def test_truncated_block_is_refused_before_anything_runs(eng):
    """A layout-2 block whose ploidy bytes demand more packed values than it holds (truncated file): the submit call is
    refused on the host — the decode kernels, which index the values from the ploidy bytes alone, never see it — and the
    stream goes on."""
    import rvtests_amd
    rng = np.random.default_rng(13)
    N = 3000
    X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=4)
    eng.fit_null(0, X, y)
    blocks = [bgengen.layout2_block(rng, N, 16, missing=0.01) for _ in range(4)]
    eng.submit_gene_bgen(0, blocks, 2, want_af=False)
    (clean,) = eng.collect()
    cut = list(blocks)
    cut[2] = blocks[2][:len(blocks[2]) // 2]
    for want_af in (False, True):
        with pytest.raises(rvtests_amd.RvtError):
            eng.submit_gene_bgen(1, cut, 2, want_af=want_af)
    eng.submit_gene_bgen(2, blocks, 2, want_af=False)
    (again,) = eng.collect()
    assert again.gene_id == 2 and again.skat_p == clean.skat_p and again.skato_p == clean.skato_p


def test_multi_allelic_mode_only_knows_the_first_alternative_allele(eng):
    """getGenotypeForAltAllele (src/BGenGenotypeExtractor.cpp:470-482): alt > 1 -> every sample missing, alt = 1 -> getGenotype."""
    rng = np.random.default_rng(9)
    N = 1500
    blocks = [bgengen.layout2_block(rng, N, 8, K=3, missing=0.01) for _ in range(3)]
    eng.vcf_set_alt_alleles([1, 2, 0])
    got = eng.bgen_decode(blocks, 2, N)
    want = _oracle_matrix(blocks, 2, N)
    want[:, 1] = -9.0
    assert np.array_equal(got, want)
    assert np.array_equal(eng.bgen_decode(blocks, 2, N), _oracle_matrix(blocks, 2, N))     # consumed
