"""Generates tests/golden/vcf_genotype.json from the REFERENCE's own libVcf parsers (oracle/_ref/libref_vcf.so, built by
`make -C oracle ref` from /root/reference/libVcf/VCFIndividual.cpp, VCFValue.cpp and base/Utils.cpp): the genotype code
VCFIndividual::parse + justGet(idx) + VCFValue::getGenotype give for every column of up to 4 bytes over the alphabet
"012./|-A:" (columns ending in ':' excluded: the reference's parser does not terminate cleanly on them), for the FORMAT
indices 0, 1, 2.  Codes: '0' '1' '2', 'm' = MISSING_GENOTYPE (-9).  Run from the repository root in the build container."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import orc  # noqa: E402
from test_vcf_cpu import ALPHABET, enumerate_columns  # noqa: E402

R = orc.ref_vcf()
assert R is not None, "reference tree needed"
cols = list(enumerate_columns(4))
codes = {}
for idx in (0, 1, 2):
    s = []
    for col in cols:
        c = R.ref_vcf_column_genotype(col, len(col), idx)
        s.append("m" if c < 0 else str(c))
    codes[str(idx)] = "".join(s)
alt_codes = {}
for alt in (1, 2):                       # VCFValue::countAltAllele(alt) of subfield 0 (multi-allelic mode)
    s = []
    for col in cols:
        c = R.ref_vcf_column_alt(col, len(col), 0, alt)
        s.append("m" if c < 0 else str(c))
    alt_codes[str(alt)] = "".join(s)
male_codes = {}                          # hemizygous regions, males: getMaleNonParGenotype02 ("0") and
for alt in (0, 1, 2):                    # countMaleNonParAltAllele2(alt) ("1", "2"; release build: asserts compiled out)
    s = []
    for col in cols:
        c = R.ref_vcf_column_male02(col, len(col), 0) if alt == 0 else R.ref_vcf_column_male_alt(col, len(col), 0, alt)
        s.append("m" if c < 0 else str(c))
    male_codes[str(alt)] = "".join(s)
out = {"alphabet": ALPHABET.decode(), "max_len": 4, "n_columns": len(cols), "codes": codes, "alt_codes": alt_codes,
       "male_codes": male_codes,
       "source": "libVcf/VCFIndividual.h:27-58,88-93 + libVcf/VCFValue.h:74-117 compiled from /root/reference"}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vcf_genotype.json")
json.dump(out, open(path, "w"))
print("wrote", path, len(cols), "columns")
