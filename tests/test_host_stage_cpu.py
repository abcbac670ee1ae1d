"""CPU: the host side of the staged host-to-device copies (rvtests_amd/csrc/host_stage.h) — the thread pool that copies the
caller's pageable memory into pinned chunks and the chunking of 1-D / 2-D copies — through the host test harness, with a
fake device (plain memory).  Whether the pool SCALES depends on the host's memory system; the scaling assertion proper runs
on the GPU box (test_gpu_stream.py), here the rates are only required not to collapse."""
import ctypes as C

import pytest

import hc


def _lib():
    L = hc.lib()
    L.hc_copy_rate.restype = C.c_double
    L.hc_copy_rate.argtypes = [C.c_size_t, C.c_int, C.c_int]
    L.hc_stage_copy2d.restype = C.c_int
    L.hc_stage_copy2d.argtypes = [C.c_size_t] * 5 + [C.c_int, C.c_int]
    return L


@pytest.mark.parametrize("width,rows,spitch,dpitch,chunk,chunks,threads", [
    (1000, 37, 1100, 1024, 8192, 3, 4),            # several whole rows per chunk, padded pitches on both sides
    (1000, 37, 1000, 1000, 8192, 2, 1),            # contiguous source
    (100000, 5, 100000, 100016, 8192, 2, 4),       # a row longer than a chunk: split into pieces
    (4000000, 3, 4000000, 4000128, 1 << 20, 4, 8),  # the shape of a genotype column (N = 500 000 doubles)
    (8, 1, 8, 16, 4096, 1, 2),                     # one tiny row, one chunk
    (4096, 1000, 4096, 4096, 4096, 4, 3),          # exactly one row per chunk
])
def test_staged_copy_puts_every_byte_where_memcpy2d_would(width, rows, spitch, dpitch, chunk, chunks, threads):
    assert _lib().hc_stage_copy2d(width, rows, spitch, dpitch, chunk, chunks, threads) == 0


def test_copy_pool_rates():
    L = _lib()
    r1 = L.hc_copy_rate(64 << 20, 1, 3)
    r4 = L.hc_copy_rate(64 << 20, 4, 3)
    print("copy pool: 1 thread %.1f GB/s, 4 threads %.1f GB/s" % (r1, r4))
    assert r1 > 0.2 and r4 > 0.2 * r1          # (a loaded CI host: the test is about collapse, not speed)


@pytest.mark.parametrize("n,max_piece,chunk,chunks,threads,seed", [
    (50, 2 << 20, 32 << 20, 4, 3, 1),      # a gene's VCF records: 2 MB pieces, 32 MB chunks
    (7, 100, 4096, 2, 1, 2),               # tiny pieces, one chunk
    (200, 70000, 1 << 20, 3, 4, 3),        # many pieces per chunk, several chunks
    (5, 3 << 20, 1 << 20, 4, 2, 4),        # pieces longer than a chunk
    (1, 10, 64, 1, 1, 5),
    (40, 5000, 4096, 2, 3, 6),             # pieces around the chunk size: both paths mixed
    (30, 9000, 4096, 3, 2, 7),
])
def test_gathered_copy_assembles_the_device_range(n, max_piece, chunk, chunks, threads, seed):
    """StageRing::copy_gather: every piece at its offset, zeros in the gaps and behind the last piece, nothing beyond."""
    L = _lib()
    L.hc_stage_gather.restype = C.c_int
    L.hc_stage_gather.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_uint]
    assert L.hc_stage_gather(n, max_piece, chunk, chunks, threads, seed) == 0
