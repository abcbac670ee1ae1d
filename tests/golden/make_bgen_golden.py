"""Generates tests/golden/bgen_blocks.json from the REFERENCE's own BGEN test data (/root/reference/libBgen/test): the
uncompressed genotype-probability blocks of selected variants of example.v11.bgen (layout 1), example.16bits.bgen
(layout 2, 16 bits), complex.bgen / complex.1bits.bgen / complex.31bits.bgen (layout 2: mixed ploidy, phased and
unphased, 1..8 alleles, 1- and 31-bit probabilities) together with the probabilities the reference's `testBGenFile`
prints for them (libBgen/test/*.vcf.correct; `%g` of BGenVariant::printGP / printHP).  The container (header, variant
identifying data, zlib) is parsed here with Python; only data travels: blocks and expected strings.  Run from the
repository root in the build container."""
import base64
import json
import os
import struct
import zlib

REF = "/root/reference/libBgen/test"


def parse(path):
    b = open(path, "rb").read()
    off, = struct.unpack_from("<I", b, 0)
    LH, M, N = struct.unpack_from("<III", b, 4)
    flags, = struct.unpack_from("<I", b, 4 + LH - 4)
    comp, layout = flags & 3, (flags >> 2) & 15
    p = off + 4
    out = []

    def s(lenbytes):
        nonlocal p
        L, = struct.unpack_from("<H" if lenbytes == 2 else "<I", b, p)
        p += lenbytes
        r = b[p:p + L]
        p += L
        return r

    for _ in range(M):
        if layout == 1:
            p += 4
        s(2), s(2), s(2)
        pos, = struct.unpack_from("<I", b, p)
        p += 4
        K = 2
        if layout == 2:
            K, = struct.unpack_from("<H", b, p)
            p += 2
        for _k in range(K):
            s(4)
        C, = struct.unpack_from("<I", b, p)
        p += 4
        if layout == 1:
            data = b[p:p + C]
            p += C
            blk = zlib.decompress(data) if comp == 1 else data
        elif comp:
            p += 4
            data = b[p:p + C - 4]
            p += C - 4
            assert comp == 1
            blk = zlib.decompress(data)
        else:
            blk = b[p:p + C]
            p += C
        out.append({"pos": pos, "K": K, "block": blk})
    return N, layout, out


def expected(path):
    rows = []
    for line in open(path):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        rows.append([c.split(":", 1)[1] for c in f[9:]])       # the probability part of GT:GP / GT:HP
    return rows


cases = []
for name, correct, take in (("example.v11", "example.v11", 10), ("example.16bits", "example.16bits.zstd", 10),
                            ("complex", "complex", None), ("complex.1bits", "complex.1bits", None),
                            ("complex.31bits", "complex.1bits", None)):
    N, layout, var = parse(os.path.join(REF, name + ".bgen"))
    exp = expected(os.path.join(REF, correct + ".bgen.vcf.correct"))
    assert len(exp) == len(var)
    idx = range(len(var)) if take is None else [0, 1, 2, 5, 17, 60, 99, 150, 197, 198][:take]
    for j in idx:
        cases.append({"file": name + ".bgen", "variant": j, "layout": layout, "N": N, "K": var[j]["K"],
                      "block": base64.b64encode(zlib.compress(var[j]["block"], 9)).decode(), "probs": exp[j]})
out = {"source": "libBgen/test/*.bgen and *.vcf.correct of /root/reference (block = base64(zlib(uncompressed block)))",
       "cases": cases}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bgen_blocks.json")
json.dump(out, open(path, "w"))
print("wrote", path, len(cases), "variants", os.path.getsize(path), "bytes")
