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


def test_malformed_record_voids_its_own_gene_only(eng):
    """A record with a wrong column count in a STREAMED gene (nobody waits for its allele frequencies): that gene's record
    comes back with RVT_ST_INPUT_ERROR and every test NA — also when it is the last gene before collect — and the genes
    around it are the ones the clean stream gives.  With want_af the submit call itself fails."""
    import rvtests_amd
    N, d, n_file = 3000, 2, 3000
    rng = np.random.default_rng(77)
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=4)
    eng.fit_null(0, X, y)
    eng.vcf_set_samples(np.arange(n_file, dtype=np.int32))
    genes = [[vcfgen.make_record(rng, n_file, pos=100 * g + j) for j in range(6)] for g in range(5)]
    for g, lines in enumerate(genes):
        eng.submit_gene_vcf(g, lines, want_af=False)
    clean = eng.collect()
    bad = [list(l) for l in genes]
    bad[1][3] = vcfgen.make_record(rng, n_file - 1, pos=777)       # one column short
    bad[4][0] = vcfgen.make_record(rng, n_file + 2, pos=778)       # the LAST gene before collect
    for g, lines in enumerate(bad):
        eng.submit_gene_vcf(g, lines, want_af=False)
    got = eng.collect()
    ST_INPUT_ERROR = 16
    assert [r.gene_id for r in got] == [0, 1, 2, 3, 4]
    for g, (r, c) in enumerate(zip(got, clean)):
        if g in (1, 4):
            assert r.status & ST_INPUT_ERROR and not (r.skat_ok or r.skato_ok or r.cmc_ok or r.zeg_ok)
        else:
            assert r.status == c.status and r.skat_p == c.skat_p and r.skato_p == c.skato_p and r.cmc_p == c.cmc_p
    assert b"gene 4" in eng.L.rvt_last_error(eng.ctx)
    with pytest.raises(rvtests_amd.RvtError):                       # somebody waits: the call itself is refused
        eng.submit_gene_vcf(9, bad[1], want_af=True)
    eng.submit_gene_vcf(10, genes[2], want_af=True)                 # ... and nothing sticks
    (r,) = eng.collect()
    assert r.gene_id == 10 and r.skat_p == clean[2].skat_p


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


def _dosage_record(rng, n_file, fmt=b"GT:DS", pos=1):
    keys = fmt.split(b":")
    pool = [b"0", b"1", b"2", b".", b"0.5", b"1.25", b"0.001", b"1e-2", b"2.000", b"1.999999", b"0.333333333333333",
            b"-0", b"+1.5", b" 0.75", b"1.5e0", b".25", b"3.", b"0.1234567", b"12345.678e-4", b"abc", b""]
    cols = []
    for s_ in range(n_file):
        parts = []
        for k in keys:
            if k == b"DS":
                parts.append(pool[rng.integers(len(pool))] if rng.random() < 0.3 else (b"%.3f" % rng.uniform(0, 2)))
            elif k == b"GT":
                parts.append([b"0/0", b"0/1", b"1/1", b"./."][rng.integers(4)])
            else:
                parts.append(b"%d" % rng.integers(0, 60))
        if rng.random() < 0.03:
            parts = parts[:1]                               # truncated column: the tag's subfield is absent -> 0.0
        if parts[-1] == b"":
            parts[-1] = b"0"                                # (no empty trailing subfield: malformed for the reference too)
        cols.append(b":".join(parts))
    head = b"\t".join([b"1", b"%d" % pos, b".", b"A", b"G", b"50", b"PASS", b".", fmt])
    return head + b"\t" + b"\t".join(cols)


@pytest.mark.parametrize("fmt,filters", [(b"GT:DS", (0, 0, 0, 0)), (b"DS", (0, 0, 0, 0)), (b"GT:GD:DS:GQ", (10, 0, 20, 0))])
def test_dosage_decode_matches_atof_bit_exactly(eng, fmt, filters):
    """--dosage TAG: the device's decimal parser against atof (the oracle calls libc), value by value, bit for bit."""
    import rvtests_amd.engine as e
    rng = np.random.default_rng(len(fmt))
    n_file, n_keep = 4000, 3500
    rows = _sample_map(rng, n_file, n_keep)
    lines = [_dosage_record(rng, n_file, fmt, pos=5 + j) for j in range(5)]
    eng.vcf_set_samples(rows)
    eng.vcf_set_filters(*filters)
    got = eng.vcf_decode_dosage(lines, b"DS", n_keep)
    for j, ln in enumerate(lines):
        off, gt, gd, gq = e.vcf_locate(eng.L, ln)
        tag = orc.vcf_format_index(fmt, b"DS")
        want, n = orc.vcf_decode_record_dosage(ln[off:], rows, n_keep, tag, gd, gq, filters)
        assert n == n_file
        assert (got[:, j].view(np.int64) == want.view(np.int64)).all()      # bit patterns (also -0.0)


def test_dosage_the_device_cannot_round_is_an_error(eng):
    import rvtests_amd
    eng.vcf_set_samples(np.arange(3, dtype=np.int32))
    eng.vcf_set_filters(0, 0, 0, 0)
    head = b"1\t5\t.\tA\tG\t50\tPASS\t.\tDS\t"
    with pytest.raises(rvtests_amd.RvtError):
        eng.vcf_decode_dosage([head + b"0.12345678901234567\t1\t2"], b"DS", 3)
    assert eng.vcf_decode_dosage([head + b"0.5\t1\t2"], b"DS", 3)[:, 0].tolist() == [0.5, 1.0, 2.0]


def test_submit_gene_vcf_dosage_equals_raw_hand_off(eng):
    N, d, n_file = 3000, 2, 3200
    rng = np.random.default_rng(77)
    rows = _sample_map(rng, n_file, N)
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=14)
    eng.fit_null(0, X, y)
    eng.vcf_set_samples(rows)
    eng.vcf_set_filters(0, 0, 0, 0)
    eng.vcf_set_dosage(True)
    lines = [_dosage_record(rng, n_file, b"GT:DS", pos=j) for j in range(14)]
    af_v = eng.submit_gene_vcf_dosage(0, lines, b"DS")
    rec_v = eng.collect()[0]
    eng.vcf_set_dosage(False)
    import rvtests_amd.engine as e
    Graw = np.zeros((N, len(lines)), order="F")
    for j, ln in enumerate(lines):
        off, gt, gd, gq = e.vcf_locate(eng.L, ln)
        Graw[:, j] = orc.vcf_decode_record_dosage(ln[off:], rows, N, 1)[0]
    af_r = eng.submit_gene_raw(0, Graw)
    rec_r = eng.collect()[0]
    assert (af_v == af_r).all()
    # (the raw doubles start on the hard-call kernel and are handed back: same fp64 sufficient statistics, but the burden
    # sums come from burden_fallback_kernel instead of the mask planes — another summation order)
    for f in ("status", "n_poly", "skat_Q", "skat_p", "skato_Q", "skato_p"):
        assert getattr(rec_v, f) == getattr(rec_r, f), f
    for f in ("cmc_p", "zeg_p"):
        assert abs(getattr(rec_v, f) - getattr(rec_r, f)) <= 1e-12 * abs(getattr(rec_r, f)), f


def test_multi_allelic_mode_counts_the_requested_allele(eng):
    """rvt_vcf_set_alt_alleles: record j counts alternative allele alt[j] (countAltAllele); consumed by one call."""
    rng = np.random.default_rng(12)
    n_file = 2000
    pool = [b"0/0", b"0/1", b"1/2", b"2/2", b"0|2", b"3/1", b"./.", b".", b"2", b"1", b"2/.", b"1/2/3", b"x/2", b"2/x"]
    recs, alts = [], [1, 2, 3, 0, 2]
    head = b"\t".join([b"1", b"77", b".", b"A", b"C,G,T", b"50", b"PASS", b".", b"GT:DP"])
    for j in range(len(alts)):
        cols = [pool[rng.integers(len(pool))] + b":%d" % rng.integers(1, 40) for _ in range(n_file)]
        recs.append(head + b"\t" + b"\t".join(cols))
    rows = np.arange(n_file, dtype=np.int32)
    eng.vcf_set_samples(rows)
    eng.vcf_set_filters(0, 0, 0, 0)
    eng.vcf_set_alt_alleles(alts)
    got = eng.vcf_decode(recs, n_file)
    import rvtests_amd.engine as e
    for j, (ln, alt) in enumerate(zip(recs, alts)):
        off = e.vcf_locate(eng.L, ln)[0]
        cols = ln[off:].split(b"\t")
        want = [orc.vcf_column_alt(c_, 0, alt) if alt > 0 else orc.vcf_column_genotype(c_, 0) for c_ in cols]
        assert got[:, j].tolist() == want
    again = eng.vcf_decode(recs[:1], n_file)                 # the setting was consumed: bi-allelic coding again
    assert again[:, 0].tolist() == [orc.vcf_column_genotype(c_, 0) for c_ in recs[0][e.vcf_locate(eng.L, recs[0])[0]:].split(b"\t")]


def test_hemizygous_records_recode_males(eng):
    """rvt_vcf_set_sex / rvt_vcf_set_hemi: in a hemizygous record males go through getMaleNonParGenotype02 (or
    countMaleNonParAltAllele2 in multi-allelic mode), females through the ordinary rule, unknown sex is missing; the GD /
    GQ filters apply afterwards; records not flagged are untouched; the flags are consumed by one call."""
    rng = np.random.default_rng(31)
    n_file, n_keep = 4000, 3100
    rows = _sample_map(rng, n_file, n_keep)
    sex = rng.choice(np.array([1, 2, 0, 9], dtype=np.int8), size=n_file, p=[0.45, 0.45, 0.05, 0.05])
    fmts = [b"GT", b"GT:GD:GQ", b"DP:GT", b"GT"]
    lines = [vcfgen.make_record(rng, n_file, fmt=fmts[j % 4], edge=0.3, chrom=b"X", pos=5000 + j) for j in range(6)]
    hemi = [1, 1, 0, 1, 1, 0]
    alts = [0, 0, 0, 1, 2, 1]
    filters = (0, 0, 10, 0)
    eng.vcf_set_samples(rows)
    eng.vcf_set_sex(sex)
    eng.vcf_set_filters(*filters)
    eng.vcf_set_hemi(hemi)
    eng.vcf_set_alt_alleles(alts)
    got = eng.vcf_decode(lines, n_keep)
    import rvtests_amd.engine as e
    for j, ln in enumerate(lines):
        off, gt, gd, gq = e.vcf_locate(eng.L, ln)
        want, n = orc.vcf_decode_record_sex(ln[off:], rows, n_keep, gt, gd, gq, filters, alt=alts[j], hemi=hemi[j], sex=sex)
        assert n == n_file and got[:, j].tolist() == want.tolist(), j
    plain = eng.vcf_decode(lines[:2], n_keep)                 # consumed: the same records without flags
    for j in range(2):
        off, gt, gd, gq = e.vcf_locate(eng.L, lines[j])
        want, _ = orc.vcf_decode_record(lines[j][off:], rows, n_keep, gt, gd, gq, filters)
        assert plain[:, j].tolist() == want.tolist()
    assert (got[:, 1] != plain[:, 1]).any() and (got[:, 1] >= 0).any()
    # dosage mode: a male's value is doubled in a hemizygous record
    vals = rng.choice([b"0", b"0.5", b"1", b"0.125", b"1e-1", b"."], size=n_file)
    head = b"\t".join([b"X", b"9", b".", b"A", b"G", b"50", b"PASS", b".", b"DS"])
    rec = head + b"\t" + b"\t".join(vals.tolist())
    eng.vcf_set_filters(0, 0, 0, 0)
    eng.vcf_set_dosage(True)
    eng.vcf_set_hemi([1])
    d = eng.vcf_decode_dosage([rec], b"DS", n_keep)
    eng.vcf_set_dosage(False)
    off = e.vcf_locate(eng.L, rec)[0]
    want, _ = orc.vcf_decode_record_dosage_sex(rec[off:], rows, n_keep, 0, hemi=1, sex=sex)
    assert np.array_equal(d[:, 0], want)
