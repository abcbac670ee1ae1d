"""GPU: the VCF text front end (rvt_vcf_decode / rvt_submit_gene_vcf) against the oracle's restatement of the reference's
per-sample loop — genotype bytes bit-exact, and the tests' records identical to those of the int8 hand-off fed with the
oracle-decoded matrix."""
import numpy as np
import pytest

import orc
import synth
import vcfgen

pytestmark = pytest.mark.gpu


@pytest.fixture
def eng():
    import rvtests_amd
    e = rvtests_amd.Engine(0)
    yield e
    e.close()


def _sample_map(rng, n_file, n_keep):
    rows = np.full(n_file, -1, dtype=np.int32)
    keep = rng.choice(n_file, n_keep, replace=False)
    rows[keep] = rng.permutation(n_keep)
    return rows


def _oracle_matrix(lines, rows, n_rows, L, filters=(0, 0, 0, 0)):
    import rvtests_amd.engine as e
    out = np.zeros((n_rows, len(lines)), dtype=np.int8, order="F")
    for j, ln in enumerate(lines):
        off, gt, gd, gq = e.vcf_locate(L, ln)
        col, n = orc.vcf_decode_record(ln[off:], rows, n_rows, gt, gd, gq, filters)
        assert n == len(rows)
        out[:, j] = col
    return out


@pytest.mark.parametrize("n_file,n_keep,fmts,filters", [
    (300, 300, [b"GT"], (0, 0, 0, 0)),
    (5000, 4321, [b"GT", b"GT:DP:GQ", b"DP:GT", b"GD:GT:GQ", b"GTX:GT"], (0, 0, 0, 0)),
    (9000, 7000, [b"GT:GD:GQ", b"GD:GQ:GT", b"GT:GQ"], (8, 50, 15, 0)),
    (2500, 2500, [b"DP:GQ"], (0, 0, 0, 0)),                      # no GT key: everything missing
])
def test_decode_matches_oracle_bit_exactly(eng, n_file, n_keep, fmts, filters):
    rng = np.random.default_rng(n_file)
    rows = _sample_map(rng, n_file, n_keep)
    lines = [vcfgen.make_record(rng, n_file, fmt=fmts[j % len(fmts)], edge=0.15, pos=100 + j) for j in range(7)]
    eng.vcf_set_samples(rows)
    eng.vcf_set_filters(*filters)
    got = eng.vcf_decode(lines, n_keep)
    want = _oracle_matrix(lines, rows, n_keep, eng.L, filters)
    assert got.dtype == np.int8 and (got == want).all()
    assert (want >= 0).any() or fmts == [b"DP:GQ"]


def test_long_records_cross_many_segments(eng):
    """One record of 400 000 fixed-width columns (1.6 MB of text, ~390 segments): tab ordinals across segment and
    scan-chunk boundaries."""
    rng = np.random.default_rng(5)
    n_file = 400000
    codes = rng.choice([0, 1, 2, -9], size=(n_file, 2), p=[0.7, 0.2, 0.08, 0.02]).astype(np.int64)
    lines = [vcfgen.fixed_width_record(codes[:, j], pos=7 + j) for j in range(2)]
    rows = np.arange(n_file, dtype=np.int32)
    eng.vcf_set_samples(rows)
    got = eng.vcf_decode(lines, n_file)
    assert (got == codes.astype(np.int8)).all()


def test_wrong_column_count_is_reported(eng):
    import rvtests_amd
    rng = np.random.default_rng(2)
    eng.vcf_set_samples(np.arange(50, dtype=np.int32))
    bad = vcfgen.make_record(rng, 49)
    with pytest.raises(rvtests_amd.RvtError):
        eng.vcf_decode([bad], 50)
    good = vcfgen.make_record(rng, 50)
    assert eng.vcf_decode([good], 50).shape == (50, 1)          # the flag does not stick


@pytest.mark.parametrize("binary", [0, 1])
def test_submit_gene_vcf_equals_int8_hand_off(eng, binary):
    N, d = 6000, 3
    rng = np.random.default_rng(40 + binary)
    n_file = 6500
    rows = _sample_map(rng, n_file, N)
    X, y, res, v, s2 = synth.make_null(N, d, binary, seed=4)
    eng.fit_null(binary, X, y)
    eng.vcf_set_samples(rows)
    genes = []
    for g, M in enumerate((12, 33, 5)):
        genes.append([vcfgen.make_record(rng, n_file, fmt=[b"GT", b"GT:DP"][j & 1], edge=0.03, pos=1000 * g + j)
                      for j in range(M)])
    af_vcf = [eng.submit_gene_vcf(g, lines) for g, lines in enumerate(genes)]
    rec_vcf = eng.collect()
    af_i8 = []
    for g, lines in enumerate(genes):
        G8 = _oracle_matrix(lines, rows, N, eng.L)
        af_i8.append(eng.submit_gene_raw(g, G8))
    rec_i8 = eng.collect()
    for a, b in zip(af_vcf, af_i8):
        assert (a == b).all()
    fields = ("status", "n_poly", "skat_Q", "skat_p", "skato_Q", "skato_rho", "skato_p", "cmc_nonref", "cmc_p", "zeg_p")
    for r, s in zip(rec_vcf, rec_i8):
        for f in fields:
            assert getattr(r, f) == getattr(s, f), f
    # and against the oracle on the consolidated matrix of the first gene
    G8 = _oracle_matrix(genes[0], rows, N, eng.L).astype(np.float64)
    G8[G8 < 0] = np.nan
    af = np.nansum(G8, axis=0) * 0.5 / N                            # GenotypeCounter: missing in the denominator
    assert np.allclose(af_vcf[0], af, rtol=0, atol=1e-15)
