"""CPU: the BGEN oracle (oracle/orc_bgen.cpp) against the reference's own golden files (libBgen/test/*.bgen with the
probabilities its testBGenFile prints, tests/golden/bgen_blocks.json) and the genotype rule of
BGenGenotypeExtractor::getGenotype on hand-made cases."""
import base64
import json
import os
import zlib

import numpy as np

import bgengen
import orc

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bgen_blocks.json")))


def _blocks():
    for c in GOLD["cases"]:
        yield c, zlib.decompress(base64.b64decode(c["block"]))


def test_probabilities_match_the_reference_golden_files():
    n_checked = 0
    for c, blk in _blocks():
        info, missing, ploidy, index, prob = orc.bgen_decode(blk, c["layout"], c["N"])
        assert info["K"] == c["K"]
        for i in range(c["N"]):
            stored = prob[index[i]:index[i + 1]]
            if missing[i]:
                want = ",".join(["."] * len(stored))
            elif c["K"] == 1 and not info["phased"]:
                want = "1"                                              # printGPAllele1
            else:
                want = ",".join("%g" % float(x) for x in stored)
            assert want == c["probs"][i], (c["file"], c["variant"], i, want, c["probs"][i])
            n_checked += 1
    assert n_checked > 10000


def test_one_and_thirty_one_bit_files_decode_alike():
    one = {c["variant"]: c for c in GOLD["cases"] if c["file"] == "complex.1bits.bgen"}
    big = {c["variant"]: c for c in GOLD["cases"] if c["file"] == "complex.31bits.bgen"}
    assert one and set(one) == set(big)
    for j in one:
        a = orc.bgen_block_genotypes(zlib.decompress(base64.b64decode(one[j]["block"])), 2, one[j]["N"])
        b = orc.bgen_block_genotypes(zlib.decompress(base64.b64decode(big[j]["block"])), 2, big[j]["N"])
        assert np.array_equal(a, b)


def test_genotype_rule():
    # layout 1: dosage = p1 + 2 p2 in double from float probabilities; all-zero triple = missing
    v = np.array([[0, 32768, 0], [0, 0, 32768], [16384, 8192, 8192], [0, 0, 0], [32768, 0, 0]], dtype="<u2")
    g = orc.bgen_block_genotypes(v.tobytes(), 1, 5)
    assert list(g) == [1.0, 2.0, 0.75, -9.0, 0.0]
    rng = np.random.default_rng(3)
    # layout 2, unphased diploid biallelic, 8 bits: p0 = a/255, p1 = b/255, p2 = 1 - p0 - p1 (float, in order)
    blk = bgengen.layout2_block(rng, 200, 8, missing=0.1)
    info, missing, ploidy, index, prob = orc.bgen_decode(blk, 2, 200)
    g = orc.bgen_block_genotypes(blk, 2, 200)
    raw = np.frombuffer(blk[8 + 200 + 2:], dtype=np.uint8).reshape(200, 2)
    sc = np.float32(1.0 / np.float32(255.0))
    for i in range(200):
        p0, p1 = np.float32(raw[i, 0]) * sc, np.float32(raw[i, 1]) * sc
        p2 = np.float32(np.float32(np.float32(1.0) - p0) - p1)
        want = -9.0 if missing[i] else float(p1) + float(p2) * 2.0
        assert g[i] == want
    # phased diploid: the reference's rule reads stored[1] = 1 - p(hap 1) and stored[2] = p(hap 2)
    blk = bgengen.layout2_block(rng, 50, 16, phased=True, missing=0.0)
    info, missing, ploidy, index, prob = orc.bgen_decode(blk, 2, 50)
    assert info["phased"] == 1 and np.all(np.diff(index) == 4)
    g = orc.bgen_block_genotypes(blk, 2, 50)
    assert np.array_equal(g, prob[1::4].astype(np.float64) + prob[2::4].astype(np.float64) * 2.0)
    # K = 1: every non-missing sample is 2; K = 3: normalised by the first three stored probabilities
    blk = bgengen.layout2_block(rng, 40, 8, K=1, missing=0.2)
    info, missing, ploidy, index, prob = orc.bgen_decode(blk, 2, 40)
    assert np.array_equal(orc.bgen_block_genotypes(blk, 2, 40), np.where(missing != 0, -9.0, 2.0))
    blk = bgengen.layout2_block(rng, 40, 12, K=3, missing=0.0)
    info, missing, ploidy, index, prob = orc.bgen_decode(blk, 2, 40)
    g = orc.bgen_block_genotypes(blk, 2, 40)
    for i in range(40):
        b = index[i]
        tot = float(np.float32(np.float32(prob[b] + prob[b + 1]) + prob[b + 2]))
        want = (float(prob[b + 1]) + float(prob[b + 2]) * 2.0) / tot if tot > 0 else -9.0
        assert g[i] == want
    # ploidy other than 1 / 2 -> missing; a haploid biallelic sample reads into the next sample's probabilities
    blk = bgengen.layout2_block(rng, 300, 9, missing=0.0, haploid=0.3, odd=0.2)
    info, missing, ploidy, index, prob = orc.bgen_decode(blk, 2, 300)
    g = orc.bgen_block_genotypes(blk, 2, 300)
    pad = np.concatenate([prob, np.zeros(4, dtype=np.float32)])
    for i in range(300):
        if ploidy[i] in (1, 2):
            assert g[i] == float(pad[index[i] + 1]) + float(pad[index[i] + 2]) * 2.0
        else:
            assert g[i] == -9.0
