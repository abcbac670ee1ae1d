#!/usr/bin/env python3
"""Fixture for BASELINE.json configs[0] (SURVEY §8d "config 1"): the reference's own example data as rvtests sees it for
    rvtest --inVcf example/example.vcf.gz --pheno example/pheno --setFile example/setFile --kernel skat
i.e. the 9 phenotyped samples P1..P9 (example/pheno, trait y1, no covariates) and the 3 variants of `set1 1:1-3`
(example/example.vcf, GT hard calls).  Data only: genotypes, phenotypes, sample and site names.  The reference ships no
expected output for this command, so the fixture pins the INPUT of the plumbing case; the expected numbers in the tests
come from the oracle.  Run in the build container (needs /root/reference): python tests/golden/make_example_config1.py"""
import json
import os

REF = "/root/reference/example"


def main():
    ph = [ln.split() for ln in open(os.path.join(REF, "pheno")) if ln.strip()]
    hdr, rows = ph[0], ph[1:]
    iid = [r[hdr.index("iid")] for r in rows]
    y1 = [float(r[hdr.index("y1")]) for r in rows]
    y4 = [int(r[hdr.index("y4")]) for r in rows]                  # binary trait, PLINK coding 1 / 2
    rng = [ln.split() for ln in open(os.path.join(REF, "setFile")) if ln.strip()][0]
    chrom, span = rng[1].split(":")
    lo, hi = (int(t) for t in span.split("-"))
    sites, G = [], []
    for ln in open(os.path.join(REF, "example.vcf")):
        if ln.startswith("##"):
            continue
        f = ln.rstrip("\n").split("\t")
        if ln.startswith("#"):
            cols = [f.index(s) for s in iid]
            continue
        if f[0] != chrom or not (lo <= int(f[1]) <= hi):
            continue
        sites.append("%s:%s" % (f[0], f[1]))
        col = []
        for c in cols:
            gt = f[c].split(":")[0].replace("|", "/").split("/")
            col.append(-9 if "." in gt else sum(int(a) for a in gt))
        G.append(col)
    out = {"command": "rvtest --inVcf example/example.vcf.gz --pheno example/pheno --setFile example/setFile --kernel skat",
           "set": rng[0], "samples": iid, "sites": sites, "y1": y1, "y4": y4,
           "genotype_by_variant": G}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "example_config1.json"), "w") as fo:
        json.dump(out, fo, indent=1)
    print(out)


if __name__ == "__main__":
    main()
