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


@pytest.mark.parametrize("n", [1, 15, 16, 17, 1000, 4099, 500000])
def test_fp64_packing_agrees_across_instruction_sets(n):
    """pack_column_f64 (the fp64 boundary packed to PLINK 2-bit rows by the staging threads): the AVX2 and AVX-512 forms give the
    bytes, the other value and its count of the scalar form — hard calls, a column with one other value (the imputed mean),
    -0.0 as an other value, and the refusals: a second other value, an other value outside [0, 2]."""
    import time

    import numpy as np
    L = hc.lib()
    L.hc_pack_column.restype = C.c_int
    L.hc_pack_column.argtypes = [C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_ubyte), C.c_size_t, C.c_int, C.c_int,
                                 C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
    L.hc_cpu_has.restype = C.c_int
    rng = np.random.default_rng(n)
    pitch = (n + 3) // 4 + 5

    def run(g, isa, reps=1):
        out = np.full(pitch, 0xAB, dtype=np.uint8)
        mu, has, cnt = C.c_double(), C.c_int(), C.c_longlong()
        t0 = time.perf_counter()
        ok = L.hc_pack_column(g.ctypes.data_as(C.POINTER(C.c_double)), n, out.ctypes.data_as(C.POINTER(C.c_ubyte)), pitch, isa, reps,
                              C.byref(mu), C.byref(has), C.byref(cnt))
        return ok, out, mu.value, has.value, cnt.value, (time.perf_counter() - t0) / reps

    cases = []
    g = rng.integers(0, 3, n).astype(np.float64)
    cases.append(("hard calls", g, True))
    g2 = g.copy()
    g2[rng.random(n) < 0.03] = 0.137
    cases.append(("one other value", g2, True))
    g3 = g.copy()
    g3[n // 2] = -0.0
    cases.append(("-0.0 is an other value (outside nothing: accepted as 0 <= -0.0)", g3, True))
    if n > 20:
        g4 = g2.copy()
        g4[3] = 0.137
        g4[n - 2] = 0.25
        cases.append(("two other values", g4, False))
        g5 = g.copy()
        g5[n // 3] = 2.5
        cases.append(("other value above 2", g5, False))
    for name, col, want_ok in cases:
        ref = run(col, 0)
        assert bool(ref[0]) == want_ok, name
        if want_ok:                                                # the codes, by hand
            codes = np.where(col == 0, 0, np.where(col == 1, 2, np.where(col == 2, 3, 1)))
            if name.startswith("-0.0"):
                codes[n // 2] = 1
            packed = np.zeros(pitch, dtype=np.uint8)
            for e in range(4):
                part = codes[e::4].astype(np.uint8)
                packed[:len(part)] |= part << (2 * e)
            assert np.array_equal(ref[1], packed), name
        for isa in (1, 2):
            if not L.hc_cpu_has(isa):
                continue
            got = run(col, isa)
            assert got[0] == ref[0], (name, isa)
            if want_ok:
                assert np.array_equal(got[1], ref[1]) and got[2:5] == ref[2:5], (name, isa)
    if n == 500000:                                                # rates (informative: printed with -s)
        for isa in (0, 1, 2):
            if L.hc_cpu_has(isa):
                dt = run(cases[1][1], isa, reps=20)[5]
                print("pack_column_f64 isa %d: %.1f GB/s" % (isa, 8.0 * n / dt / 1e9))


@pytest.mark.parametrize("n", [1, 3, 31, 32, 33, 100, 4099, 65536 + 7])
def test_int8_packing_gives_plink_codes(n):
    """pack_column_i8 (the int8 boundary packed to PLINK 2-bit rows by the staging threads): 0 -> 00, 1 -> 10, 2 -> 11, negative ->
    01 (missing), sample p in bits 2 (p & 3) of byte p >> 2 — the scalar and the AVX2 form against a numpy statement of the codes
    (libVcf/PlinkInputFile.h:206-209); zeros behind the row up to its pitch; a value above 2 is refused by both."""
    import numpy as np
    L = hc.lib()
    L.hc_pack_column_i8.restype = C.c_int
    L.hc_pack_column_i8.argtypes = [C.POINTER(C.c_byte), C.c_size_t, C.POINTER(C.c_ubyte), C.c_size_t, C.c_int, C.c_int]
    rng = np.random.default_rng(n)
    g = rng.integers(0, 3, n).astype(np.int8)
    g[rng.random(n) < 0.1] = -9
    if n > 2:
        g[1] = -128
    pitch = (n + 3) // 4 + 7
    code = np.where(g < 0, 1, np.where(g == 0, 0, np.where(g == 1, 2, 3))).astype(np.uint8)
    code = np.concatenate([code, np.zeros((-n) % 4, dtype=np.uint8)]).reshape(-1, 4)
    want = np.concatenate([code[:, 0] | (code[:, 1] << 2) | (code[:, 2] << 4) | (code[:, 3] << 6), np.zeros(7, dtype=np.uint8)])
    for isa in (0, 1):
        out = np.full(pitch, 0xAB, dtype=np.uint8)
        ok = L.hc_pack_column_i8(g.ctypes.data_as(C.POINTER(C.c_byte)), n, out.ctypes.data_as(C.POINTER(C.c_ubyte)), pitch, isa, 1)
        assert ok == 1 and np.array_equal(out, want), isa
    for pos in {0, n // 2, n - 1}:
        bad = g.copy()
        bad[pos] = 3
        for isa in (0, 1):
            out = np.zeros(pitch, dtype=np.uint8)
            assert L.hc_pack_column_i8(bad.ctypes.data_as(C.POINTER(C.c_byte)), n, out.ctypes.data_as(C.POINTER(C.c_ubyte)), pitch, isa,
                                       1) == 0
