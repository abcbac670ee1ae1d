// rvtests_amd — which tiles a band product computes and how its K range is cut: the pure index arithmetic of band_gemm.hip.h
// (256 x 256 tiles of the int8 / MXFP4 band) and of gemm_f64.hip.h's band mode (256 x 128 tiles), shared by the kernels, their
// host launch code and the host test harness (csrc/hostcheck.cpp; tests/test_band_tiles_cpu.py walks every (head, marker) pair
// of a band through it).  No HIP types: compiled by hipcc for the device and by g++ for the harness.
#pragma once
#include <stddef.h>
#include <algorithm>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RVT_BT_HD __host__ __device__ inline
#else
#define RVT_BT_HD inline
#endif

namespace rvt {

constexpr int kBandBT = 256;  // tile edge of the integer band: heads x markers

// column tiles of row panel rp of the integer band: the markers [256 rp, min(W, 256 rp + 256 + halo)); they start ON the
// panel's own diagonal tile
RVT_BT_HD int band_panel_tiles(int rp, int W, int halo) {
  const long long lo = (long long)rp * kBandBT;
  long long hi = lo + kBandBT + halo;
  if (hi > W) hi = W;
  return hi > lo ? (int)((hi - lo + kBandBT - 1) / kBandBT) : 0;
}
inline int band_tiles(int H, int W, int halo) {
  int n = 0;
  for (int rp = 0; rp < (H + kBandBT - 1) / kBandBT; ++rp) n += band_panel_tiles(rp, W, halo);
  return n;
}
// index of the tile that holds (head h, marker j), j - h in [0, halo], j < W, in the list the kernels enumerate (row panels in
// order, each panel's column tiles in order) — what band_finish_i32_kernel computes per head
RVT_BT_HD int band_tile_of(int h, int j, int W, int halo) {
  int t0 = 0;
  for (int rp = 0; rp < (h >> 8); ++rp) t0 += band_panel_tiles(rp, W, halo);
  return t0 + (j >> 8) - (h >> 8);
}
// K slices (a multiple of 8): the count that minimises rounds x (chunks per slice + 10), a round being the 32 workgroups an
// XCD holds at once (one workgroup of 128 KB LDS per CU) and 10 chunks what a workgroup spends besides its K loop (pipeline
// fill, the 256 KB partial tile it writes and band_finish reads again) — fitted on N = 500 000: 6 / 20 / 52 tiles run fastest
// with 32-40 / 24 / 24 slices, 64 slices cost 10-20 % more; slices of at least 16 chunks; the partial tiles of all slices must
// fit `max_part_bytes`
inline long long band_slices(int n_tiles, long long chunks, size_t max_part_bytes) {
  long long best = 8, best_cost = -1;
  for (long long k = 1; k <= 16; ++k) {
    if (k > 1 && chunks / (8 * k) < 16) break;
    if (k > 1 && (size_t)n_tiles * (size_t)(8 * k) * (size_t)kBandBT * kBandBT * sizeof(int) > max_part_bytes) break;
    const long long rounds = ((long long)n_tiles * k + 31) / 32, cost = rounds * ((chunks + 8 * k - 1) / (8 * k) + 10);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = 8 * k;
    }
  }
  return best;
}

// ---- the fp64 product's tiles (gemm_f64.hip.h): 256 rows x 128 columns --------------------------------------------------------
constexpr int kGemmTileM = 256, kGemmTileN = 128;
// (halo >= 0, symmetric only: a BAND — row m needs the columns m .. m + halo; row panel rp then ends at the column tile that
//  holds column rp BM + BM - 1 + halo)
RVT_BT_HD int gemm_f64_panel_last(int rp, int n_col_tiles, int halo) {
  if (halo < 0) return n_col_tiles;
  const long long l = ((long long)rp * kGemmTileM + kGemmTileM + halo + kGemmTileN - 1) / kGemmTileN;
  return l < n_col_tiles ? (int)l : n_col_tiles;
}
inline int gemm_f64_tiles(int M, int Ntot, bool symmetric, int* n_col_tiles, int halo = -1) {
  const int nrp = (M + kGemmTileM - 1) / kGemmTileM, nct = (Ntot + kGemmTileN - 1) / kGemmTileN;
  *n_col_tiles = nct;
  if (!symmetric) return nrp * nct;
  int n = 0;
  for (int rp = 0; rp < nrp; ++rp) n += std::max(0, gemm_f64_panel_last(rp, nct, halo) - (rp * kGemmTileM) / kGemmTileN);
  return n;
}

}  // namespace rvt
