"""Synthetic VCF records for the front-end tests (test infrastructure)."""
import numpy as np

# sample columns the generators draw from: ordinary calls plus every rule of VCFValue::getGenotype
GT_POOL = [b"0/0", b"0/1", b"1/0", b"1/1", b"0|0", b"0|1", b"1|1", b"./.", b".", b".|.", b"0", b"1", b"1/2", b"2/1", b"0/2",
           b"0/.", b"./1", b"0/-", b"1/+", b"0/1/1", b"10", b"0:1", b"A/0", b"0/A", b"0/", b"1|", b"00", b"0\\1", b"\xc3/0",
           b"0/\xc3"]
GT_COMMON = [b"0/0"] * 12 + [b"0/1"] * 4 + [b"1/1", b"0|1", b"./.", b"."]


def make_record(rng, n_file, fmt=b"GT", edge=0.05, chrom=b"1", pos=1000):
    """One record line (bytes, no newline).  fmt: FORMAT column; GT takes its place, other keys get integers."""
    keys = fmt.split(b":")
    cols = []
    for s in range(n_file):
        pool = GT_POOL if rng.random() < edge else GT_COMMON
        gt = pool[rng.integers(len(pool))]
        parts = []
        for k in keys:
            if k == b"GT":
                parts.append(gt)
            else:
                parts.append(b"%d" % rng.integers(0, 60) if rng.random() > 0.03 else b".")
        if len(keys) > 1 and rng.random() < 0.02:
            parts = parts[:max(1, int(rng.integers(1, len(keys))))]     # truncated column (trailing fields dropped)
        cols.append(b":".join(parts))
    head = b"\t".join([chrom, b"%d" % pos, b".", b"A", b"G", b"50", b"PASS", b"NS=3;DP=14", fmt])
    return head + b"\t" + b"\t".join(cols)


def fixed_width_record(codes, chrom=b"1", pos=1000):
    """GT-only record from integer codes (0/1/2, negative = missing) — the common case of a large cohort file."""
    lut = np.array([b"0/0", b"0/1", b"1/1", b"./."])
    c = np.where(codes < 0, 3, codes)
    head = b"\t".join([chrom, b"%d" % pos, b".", b"A", b"G", b"50", b"PASS", b".", b"GT"])
    return head + b"\t" + b"\t".join(lut[c].tolist())
