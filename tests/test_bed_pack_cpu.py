"""PLINK .bed packing used at the rvt_submit_gene_bed boundary: Engine.pack_bed against a literal restatement of the
reference's SNP-major reader (libVcf/PlinkInputFile.cpp:24-47; codes libVcf/PlinkInputFile.h:206-209:
00 -> 0, 10 -> 1, 11 -> 2, 01 -> missing = -9; sample p in bits 2(p & 3).. of byte p >> 2, a fresh byte per variant)."""
import numpy as np
import pytest

from rvtests_amd.engine import Engine


def read_like_plink_input_file(bed, N):
    M = bed.shape[0]
    mat = np.zeros((N, M))
    for m in range(M):
        for p in range(N):
            c = int(bed[m, p >> 2])
            geno = (c >> ((p & 3) << 1)) & 3
            mat[p, m] = {0: 0, 2: 1, 3: 2, 1: -9}[geno]
    return mat


@pytest.mark.parametrize("N", [1, 3, 4, 5, 17, 64, 203])
def test_pack_bed_round_trip(N):
    rng = np.random.default_rng(N)
    G = rng.integers(-1, 3, size=(N, 7)).astype(np.float64)
    G[G < 0] = -9.0
    bed = Engine.pack_bed(G)
    assert bed.shape == (7, (N + 3) // 4) and bed.dtype == np.uint8
    assert np.array_equal(read_like_plink_input_file(bed, N), G)
    if N % 4:          # pad bits of the last byte are zero (what PLINK writes)
        assert (bed[:, -1] >> (2 * (N % 4)) == 0).all()
