"""CPU: the VCF genotype text rules.  oracle/orc_vcf.cpp against the REFERENCE's own libVcf parsers compiled where they
lie (oracle/_ref/libref_vcf.so) on an exhaustive set of short columns, against the committed golden vector of the same
set (tests/golden/vcf_genotype.json, for machines without the reference tree), and the host-side record locator of the
C ABI against the oracle's restatement of VCFRecord::getFormatIndex."""
import ctypes as C
import itertools
import json
import os

import numpy as np
import pytest

import orc
import vcfgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "vcf_genotype.json")
ALPHABET = b"012./|-A:"


def enumerate_columns(max_len=4):
    """Every string over ALPHABET up to max_len bytes that is not malformed in the ways the reference itself cannot
    handle (empty column, trailing ':')."""
    for n in range(1, max_len + 1):
        for t in itertools.product(ALPHABET, repeat=n):
            if t[-1] == ord(":"):
                continue
            yield bytes(t)


def test_oracle_matches_golden_vector():
    g = json.load(open(GOLDEN))
    assert g["alphabet"] == ALPHABET.decode() and g["max_len"] == 4
    cols = list(enumerate_columns(4))
    for idx in (0, 1, 2):
        want = g["codes"][str(idx)]
        assert len(want) == len(cols)
        got = "".join("m" if (c := orc.vcf_column_genotype(col, idx)) < 0 else str(c) for col in cols)
        assert got == want


def test_alt_allele_rule_matches_golden_and_reference():
    """Multi-allelic mode: VCFValue::countAltAllele for alt = 1, 2."""
    g = json.load(open(GOLDEN))
    cols = list(enumerate_columns(4))
    R = orc.ref_vcf()
    for alt in (1, 2):
        got = "".join("m" if (c := orc.vcf_column_alt(col, 0, alt)) < 0 else str(c) for col in cols)
        assert got == g["alt_codes"][str(alt)]
        if R is not None:
            for col in list(enumerate_columns(3)) + [b"1/2/3", b"12", b"9|9", b"3/3", b"2|.", b"x/2"]:
                if col.endswith(b":"):
                    continue
                for a in (1, 2, 3, 9):
                    assert orc.vcf_column_alt(col, 0, a) == R.ref_vcf_column_alt(col, len(col), 0, a), (col, a)


def test_male_hemizygous_rules_match_golden_and_reference():
    """Hemizygous regions: VCFValue::getMaleNonParGenotype02 and countMaleNonParAltAllele2 for males; the sample loop sends
    females through the ordinary rule, unknown sex to missing, and doubles a male's dosage."""
    g = json.load(open(GOLDEN))
    cols = list(enumerate_columns(4))
    for alt in (0, 1, 2):
        got = "".join("m" if (c := (orc.vcf_column_male02(col, 0) if alt == 0 else orc.vcf_column_male_alt(col, 0, alt))) < 0
                      else str(c) for col in cols)
        assert got == g["male_codes"][str(alt)]
    R = orc.ref_vcf()
    if R is not None:
        for col in list(enumerate_columns(3)) + [b"1/1/1", b"1x1", b"12", b"9|9", b"3/3", b"2|.", b"x/2", b"1", b"2", b"A"]:
            if col.endswith(b":"):
                continue
            assert orc.vcf_column_male02(col, 0) == R.ref_vcf_column_male02(col, len(col), 0), col
            for a in (1, 2, 3, 9):
                assert orc.vcf_column_male_alt(col, 0, a) == R.ref_vcf_column_male_alt(col, len(col), 0, a), (col, a)
    text = b"0/1\t1/1\t1\t0|0\t1/1\t0/1\t1"
    rows = np.arange(7, dtype=np.int32)
    sex = np.array([1, 1, 1, 1, 2, 2, 0], dtype=np.int8)
    out, n = orc.vcf_decode_record_sex(text, rows, 7, 0, hemi=1, sex=sex)
    assert n == 7 and out.tolist() == [-9, 2, 2, 0, 2, 1, -9]      # male het -> missing, unknown sex -> missing
    out, n = orc.vcf_decode_record_sex(text, rows, 7, 0, hemi=0, sex=sex)
    assert out.tolist() == [1, 2, 1, 0, 2, 1, 1]
    d, n = orc.vcf_decode_record_dosage_sex(b"0.25\t0.5\t1.5", np.arange(3, dtype=np.int32), 3, 0, hemi=1,
                                            sex=np.array([1, 2, 0], dtype=np.int8))
    assert d.tolist() == [0.5, 0.5, 1.5]


def test_oracle_matches_compiled_reference_exhaustively():
    R = orc.ref_vcf()
    if R is None:
        pytest.skip("reference tree not present: the golden vector covers this machine")
    n = 0
    for col in enumerate_columns(5):
        for idx in (0, 1, 2):
            assert orc.vcf_column_genotype(col, idx) == R.ref_vcf_column_genotype(col, len(col), idx), (col, idx)
            n += 1
    assert n > 150000
    for col in vcfgen.GT_POOL:                       # bytes >= 0x80 compare as negative chars on the host
        assert orc.vcf_column_genotype(col, 0) == R.ref_vcf_column_genotype(col, len(col), 0), col
    # the integer subfields the GD / GQ filters read (atoi of the NUL-terminated subfield; "." when absent)
    for col, idx, want in [(b"0/1:17:40", 1, 17), (b"0/1:17:40", 2, 40), (b"0/1:17", 2, 0), (b"0/1:.:3", 1, 0),
                           (b"0/1: 7x", 1, 7), (b"0/1:-4", 1, -4)]:
        assert R.ref_vcf_column_int(col, len(col), idx) == want


def test_filters_and_record_walk():
    text = b"0/1:5:30\t1/1:20:10\t0/0\t./.:9:9\t0/1:12"
    rows = np.array([2, 0, -1, 1, 3], dtype=np.int32)
    out, n = orc.vcf_decode_record(text, rows, 4, 0)
    assert n == 5 and out.tolist() == [2, -9, 1, 1]
    out, n = orc.vcf_decode_record(text, rows, 4, 0, gd_idx=1, gq_idx=2, filters=(10, 0, 0, 0))
    assert out.tolist() == [2, -9, -9, 1]            # depth 5 < 10 -> missing; absent GD in column 3 is excluded anyway
    out, n = orc.vcf_decode_record(text, rows, 4, 0, gd_idx=1, gq_idx=2, filters=(0, 15, 0, 0))
    assert out.tolist() == [-9, -9, 1, 1]            # depth 20 > GDmax
    out, n = orc.vcf_decode_record(text, rows, 4, 0, gd_idx=1, gq_idx=2, filters=(0, 0, 20, 0))
    assert out.tolist() == [-9, -9, 1, -9]           # GQ 10 < 20; column without GQ: atoi(".") = 0 < 20
    out, n = orc.vcf_decode_record(text, rows, 4, -1)
    assert (out == -9).all()                         # no GT key in FORMAT


def _lib():
    import rvtests_amd.engine as e
    return e.load_library()


def test_locate_matches_format_index_rule():
    L = _lib()
    import rvtests_amd.engine as e
    for fmt in (b"GT", b"GT:DP:GQ", b"DP:GT", b"GD:GT:GQ", b"GTX:GT", b"DP:GDX:GQ", b"DP", b"A:B:GQ:GT:GD"):
        line = b"1\t100\trs1\tA\tG\t.\tPASS\tAC=1\t" + fmt + b"\t0/1\t1/1"
        off, gt, gd, gq = e.vcf_locate(L, line)
        assert line[off:] == b"0/1\t1/1"
        assert gt == orc.vcf_format_index(fmt, b"GT")
        assert gd == orc.vcf_format_index(fmt, b"GD")
        assert gq == orc.vcf_format_index(fmt, b"GQ")
    assert orc.vcf_format_index(b"GTX:GT", b"GT") == 0           # the reference's prefix match
    with pytest.raises(ValueError):
        e.vcf_locate(L, b"1\t100\trs1\tA\tG\t.\tPASS\tAC=1")     # no FORMAT / sample columns
