"""CPU: the tile lists of MetaCov's band products (rvtests_amd/csrc/band_tiles.h — the index arithmetic shared by
band_gemm.hip.h, gemm_f64.hip.h, their host launch code and the finishing kernels), through the host test harness.
Every (head, marker) pair the adapter prints must lie in exactly one enumerated tile, at the index the finishing kernel
computes for it, and nothing far outside the band may be enumerated (the point of the band: a heads x window rectangle
computes about twice what is printed)."""
import ctypes as C

import numpy as np
import pytest

import hc


def _lib():
    L = hc.lib()
    L.hc_band_tiles.restype = C.c_int
    L.hc_band_tiles.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.hc_band_tile_of.restype = C.c_int
    L.hc_band_tile_of.argtypes = [C.c_int] * 4
    L.hc_band_slices.restype = C.c_longlong
    L.hc_band_slices.argtypes = [C.c_int, C.c_longlong, C.c_longlong]
    L.hc_gemm_f64_tiles.restype = C.c_int
    L.hc_gemm_f64_tiles.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    return L


CASES = [(1, 1, 0), (1, 5, 4), (255, 255, 0), (256, 256, 255), (257, 300, 43), (700, 1000, 300), (1024, 2024, 1000),
         (1024, 1100, 1000), (4096, 7096, 3000), (3072, 4096, 1000), (513, 513, 512), (300, 1300, 1000), (5000, 5040, 40)]


@pytest.mark.parametrize("H,W,halo", CASES)
def test_integer_band_tile_list_covers_the_band_exactly_once(H, W, halo):
    L = _lib()
    n = L.hc_band_tiles(H, W, halo, None)
    out = (C.c_int * (3 * n))()
    assert L.hc_band_tiles(H, W, halo, out) == n
    tiles = np.array(out, dtype=np.int64).reshape(n, 3)
    assert len({(int(a), int(b)) for a, b, _ in tiles}) == n                       # no tile twice
    index = {(int(rp), int(ct)): t for t, (rp, ct, _) in enumerate(tiles)}
    # every printed pair: head h < H, marker j = h .. min(W - 1, h + halo)
    rng = np.random.default_rng(H + W + halo)
    heads = np.unique(np.concatenate([np.arange(0, H, max(1, H // 97)), [H - 1], rng.integers(0, H, 50)]))
    for h in heads:
        h = int(h)
        js = np.unique(np.concatenate([[h, min(W - 1, h + halo)], rng.integers(h, min(W - 1, h + halo) + 1, 20)]))
        for j in js:
            j = int(j)
            t = index.get((h >> 8, j >> 8))
            assert t is not None, (h, j)
            assert L.hc_band_tile_of(h, j, W, halo) == t, (h, j)                   # what band_finish_i32_kernel computes
    # nothing far from the band: a tile (rp, ct) meets the band iff some head of the panel reaches a marker of the tile
    for rp, ct, k in tiles:
        lo_h, hi_h = 256 * rp, min(H, 256 * rp + 256) - 1
        lo_j, hi_j = 256 * ct, min(W, 256 * ct + 256) - 1
        assert lo_j <= W - 1 and hi_j >= lo_h and lo_j <= 256 * rp + 255 + halo, (rp, ct)   # (panels are cut at 256, not at H)
    # the band's tiles against the rectangle's: at least the diagonal, at most ceil((256 + halo) / 256) per panel
    per_panel = np.bincount(tiles[:, 0])
    assert per_panel.max() <= (256 + halo + 255) // 256 and per_panel.min() >= 1


@pytest.mark.parametrize("M,Ntot,halo", [(300, 300, -1), (1024, 1024, -1), (900, 900, 100), (700, 700, 300), (1024, 2024, 1000),
                                         (520, 520, 0), (257, 129, -1), (4096, 7096, 3000)])
def test_fp64_band_tile_list_covers_the_band(M, Ntot, halo):
    L = _lib()
    n = L.hc_gemm_f64_tiles(M, Ntot, 1, halo, None)
    out = (C.c_int * (2 * n))()
    L.hc_gemm_f64_tiles(M, Ntot, 1, halo, out)
    tiles = {(out[2 * t], out[2 * t + 1]) for t in range(n)}
    assert len(tiles) == n
    rng = np.random.default_rng(M + Ntot)
    for m in np.unique(np.concatenate([[0, M - 1], rng.integers(0, M, 200)])):
        m = int(m)
        hi = Ntot - 1 if halo < 0 else min(Ntot - 1, m + halo)
        if hi < m:
            continue
        for j in np.unique(np.concatenate([[m, hi], rng.integers(m, hi + 1, 10)])):
            assert (m // 256, int(j) // 128) in tiles, (m, int(j))
    if halo >= 0:          # far fewer tiles than the upper triangle when the band is narrow
        full = L.hc_gemm_f64_tiles(M, Ntot, 1, -1, None)
        assert n <= full
        if Ntot > 4 * (halo + 256):
            assert n < 0.7 * full
    # not symmetric: the whole rectangle
    assert L.hc_gemm_f64_tiles(M, Ntot, 0, -1, None) == ((M + 255) // 256) * ((Ntot + 127) // 128)


def test_band_slices_heuristic():
    """A multiple of 8 (slice s goes to XCD s mod 8), at least 16 chunks per slice, within the memory given for the partial
    tiles; the fitted points of profiles/r6_band_slices_sweep.txt (N = 500 000, MXFP4: 1 954 chunks)."""
    L = _lib()
    for n_tiles, chunks in [(1, 4), (6, 1954), (20, 1954), (52, 1954), (208, 1954), (20, 3907), (500, 1954), (3, 40)]:
        for cap in (3 << 30, 64 << 20):
            s = L.hc_band_slices(n_tiles, chunks, cap)
            assert s % 8 == 0 and 8 <= s <= 128
            if s > 8:
                assert chunks // s >= 16 and n_tiles * s * 256 * 256 * 4 <= cap
    assert L.hc_band_slices(20, 1954, 3 << 30) == 24 and L.hc_band_slices(52, 1954, 3 << 30) == 24
    assert L.hc_band_slices(6, 1954, 3 << 30) in (32, 40)
