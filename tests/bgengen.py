"""Synthetic BGEN genotype-probability blocks (uncompressed, as the reference holds them after inflating) for tests and
benchmarks: layout 1 (3 x uint16 per sample) and layout 2 (ploidy / missing bytes, phased flag, B-bit packed values)."""
import math

import numpy as np


def layout1_block(rng, N, missing=0.02):
    p = rng.dirichlet([1.0, 1.0, 1.0], size=N)
    v = np.minimum(np.round(p * 32768), 65535).astype("<u2")
    v[rng.random(N) < missing] = 0
    return v.tobytes()


def n_values(Z, K, phased):
    return Z * (K - 1) if phased else math.comb(Z + K - 1, K - 1) - 1


def layout2_block(rng, N, bits, K=2, phased=False, ploidy=2, missing=0.02, haploid=0.0, odd=0.0):
    """ploidy: the common ploidy; `haploid` / `odd`: fractions of samples with ploidy 1 / 3 (or 0)."""
    Z = np.full(N, ploidy, dtype=np.int64)
    r = rng.random(N)
    Z[r < haploid] = 1
    Z[(r >= haploid) & (r < haploid + odd)] = rng.choice([0, 3], size=int(((r >= haploid) & (r < haploid + odd)).sum()))
    miss = rng.random(N) < missing
    pm = (Z | (miss.astype(np.int64) << 7)).astype(np.uint8)
    nv = np.array([n_values(int(z), K, phased) for z in Z], dtype=np.int64)
    total = int(nv.sum())
    top = (1 << bits) - 1
    # values of one sample sum to <= top (roughly): random splits of `top`
    vals = np.zeros(total, dtype=np.uint64)
    pos = 0
    for i in range(N):
        k = int(nv[i])
        if k == 0:
            continue
        if phased:
            per = K - 1
            for h in range(int(Z[i])):
                cuts = np.sort(rng.integers(0, top + 1, size=per, dtype=np.uint64))
                prev = 0
                for c in cuts:
                    vals[pos] = int(c) - prev
                    prev = int(c)
                    pos += 1
        else:
            cuts = np.sort(rng.integers(0, top + 1, size=k, dtype=np.uint64))
            prev = 0
            for c in cuts:
                vals[pos] = int(c) - prev
                prev = int(c)
                pos += 1
    # pack LSB first
    nbytes = (total * bits + 7) // 8
    acc = 0
    nacc = 0
    out = bytearray()
    for v in vals:
        acc |= int(v) << nacc
        nacc += bits
        while nacc >= 8:
            out.append(acc & 0xFF)
            acc >>= 8
            nacc -= 8
    if nacc:
        out.append(acc & 0xFF)
    assert len(out) == nbytes
    head = np.array([N], dtype="<u4").tobytes() + np.array([K], dtype="<u2").tobytes() + bytes([int(Z.min()), int(Z.max())])
    return head + pm.tobytes() + bytes([1 if phased else 0, bits]) + bytes(out)


def layout2_block_fast(rng, N, bits=16, missing=0.01):
    """Unphased diploid biallelic block with byte-aligned probabilities (8 / 16 / 32 bits), vectorised (bench sizes)."""
    assert bits in (8, 16, 32)
    top = (1 << bits) - 1
    maf = 10 ** rng.uniform(-3, -1)
    g = rng.binomial(2, maf, size=N)
    p = np.full((N, 3), 0.01)
    p[np.arange(N), g] = 0.98
    v = np.round(p[:, :2] * top).astype({8: "<u1", 16: "<u2", 32: "<u4"}[bits])
    pm = np.full(N, 2, dtype=np.uint8)
    pm[rng.random(N) < missing] |= 0x80
    head = np.array([N], dtype="<u4").tobytes() + np.array([2], dtype="<u2").tobytes() + bytes([2, 2])
    return head + pm.tobytes() + bytes([0, bits]) + v.tobytes()
