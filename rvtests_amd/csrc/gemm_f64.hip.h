// rvtests_amd — C = A' D B in fp64 on the matrix cores, for tall operands that are contiguous along the contraction index.
//
// What it serves: the score-covariance band of MetaCovTest (src/Model.cpp:844-1004; the per-pair fp32 dot products of
// N-vectors in MetaCovUnrelatedQtl / MetaCovUnrelatedBinary, :506-593, :694-778) for blocks that are NOT hard calls —
// dosages, or any block under a binary trait (D = diag(p(1-p))): S = G_H' D G_W and T = G_W' D X, where G is the N x V
// block of the window in the boundary's column-major layout (a variant's samples are contiguous: both operands of the
// product run along K = the sample index, a "TN" product).  Hard-call blocks under an unweighted model keep the exact int8
// product (rot_gemm.hip.h).
//
// Round 4 computed this band with the one-wave sufficient-statistics kernel (gene_suffstat_panel: no LDS, 64 x 64 outputs per
// wave): every column panel of a 1024-variant block was fetched ~16 times and the kernel ran at 22 TFLOP/s = 0.28 of the fp64
// matrix peak.  Here a workgroup of four waves (one per SIMD, 512 registers each: the 256 accumulator registers of a
// 128 x 64 wave tile live in AGPRs) owns a 256 x 128 tile of C; K is consumed in chunks of 16 samples = one 128-byte line per
// operand row, which go from global memory straight into LDS (global_load_lds_dwordx4, XOR-swizzled 16-byte segments exactly
// as in rot_gemm.hip.h, two or three stages with counted vmcnt and ONE barrier per chunk); a lane's ds_read_b128 holds the
// operand values of TWO k-steps (samples 2g, 2g + 1 of an 8-sample step, g = lane >> 4 — the contraction index may be
// permuted freely as long as both operands agree), so a chunk costs each wave 24 LDS reads for 128 matrix instructions of 64
// cycles.  The workgroup reads (256 + 128) x 128 B per 8192 matrix-pipe cycles: 6 B per cycle and CU, and the tiles that run
// side by side on one XCD are the tiles of ONE K slice (see the index map below), so a chunk of the block is fetched from
// HBM about once per XCD that works on its slice.
// K is split across workgroups (few output tiles: 20 for the upper triangle of 1024 x 1024); every slice writes its own
// partial C and rot_reduce_slices_kernel adds them in a fixed order — bit-reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "band_tiles.h"    // gemm_f64_tiles, gemm_f64_panel_last
#include "rot_gemm.hip.h"  // rot_lds_off, rot_wait_vm_barrier

namespace rvt {

typedef double gd4_t __attribute__((ext_vector_type(4)));
typedef double gd2_t __attribute__((ext_vector_type(2)));

constexpr int kGemmWM = 2, kGemmWN = 2, kGemmTM = 8, kGemmTN = 4;         // 4 waves, 128 x 64 outputs each
constexpr int kGemmBM = 16 * kGemmWM * kGemmTM, kGemmBN = 16 * kGemmWN * kGemmTN;   // 256 x 128
static_assert(kGemmBM == kGemmTileM && kGemmBN == kGemmTileN, "band_tiles.h enumerates 256 x 128 tiles");
constexpr int kGemmKC = 16;                                               // samples per chunk (128 bytes per row)
constexpr int kGemmThreads = 64 * kGemmWM * kGemmWN;

// A: column m of the row side starts at A + m * lda (M columns).  B side: columns [0, Nb) at B + j * ldb, then columns
// [Nb, Nb + Nb2) at B2 + (j - Nb) * ldb2 (the null-model columns behind the genotype columns).  w: optional weights along K.
// Every column must be readable (and zero) up to K rounded up to 16 samples — the engine's blocks are (ld is a multiple of
// 16, pad rows zero).  Slice s covers samples [s kslice, (s + 1) kslice) and writes C + s c_slice; C[m + j ldc].
// symmetric: A and B are the same columns — tiles entirely below the diagonal are skipped (their C entries are not written).
// n_tiles = the tiles that are computed (gemm_f64_tiles); grid = 8 * n_tiles * ceil(n_slices / 8), see the index map.
// (the tiles of an M x Ntot product that are computed: gemm_f64_tiles, band_tiles.h)
// K slices (a multiple of 8: XCD x takes every 8th slice): the count that minimises rounds x chunks per workgroup, a round
// being the 32 workgroups an XCD holds at once; slices of at least 64 chunks
inline long long gemm_f64_slices(int n_tiles, long long chunks) {
  long long best = 8, best_cost = -1;
  for (long long k = 1; k <= 8; ++k) {
    if (k > 1 && chunks / (8 * k) < 64) break;
    const long long rounds = (n_tiles * k + 31) / 32, cost = rounds * ((chunks + 8 * k - 1) / (8 * k));
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = 8 * k;
    }
  }
  return best;
}

template <int NST, bool SUB = false>
__global__ __launch_bounds__(kGemmThreads, 1) void gemm_tn_f64_kernel(
    const double* __restrict__ A, long long lda, int M, const double* __restrict__ B, long long ldb, int Nb,
    const double* __restrict__ B2, long long ldb2, int Nb2, const double* __restrict__ w, long long K, long long kslice,
    int n_slices, double* __restrict__ C0, long long ldc, long long c_slice, int n_tiles, int n_col_tiles,
    int symmetric, int halo = -1, int ring = 0, int col0 = 0) {
  // ring > 0: A and B are the base of a block used as a ring of `ring` columns; column m of either side is the physical
  // column (col0 + m) mod ring (MetaCov's circular window, band_gemm.hip.h)
  constexpr int WM = kGemmWM, WN = kGemmWN, TM = kGemmTM, TN = kGemmTN, BM = kGemmBM, BN = kGemmBN, KC = kGemmKC;
  constexpr int kWaves = WM * WN;
  constexpr int kPieces = (BM + BN) / 8, PPW = kPieces / kWaves;  // 1 KiB pieces: 8 rows x 128 B
  constexpr int kStage = (BM + BN) * 128;
  static_assert(kPieces % kWaves == 0, "pieces must divide evenly among the waves");
  static_assert(PPW * (NST - 1) < 64, "the ring's loads must fit the vmcnt counter");
  __shared__ __attribute__((aligned(1024))) char lds[NST][kStage];
  // index map: workgroup id -> (XCD = id & 7, w = id >> 3).  XCD x takes the K slices x, x + 8, ...; inside an XCD the
  // workgroups walk the tiles of one slice before the next slice, so the 32 workgroups resident on an XCD read the same
  // sample range of the block at the same time and share it through the XCD's L2.  Only the tiles that are computed are
  // enumerated (n_tiles of them): workgroups that return at once are not harmless — the workgroups of an XCD are dealt to
  // its four shader engines in turn, and with the skipped tiles of a symmetric product in the list two of the four engines
  // received 18 of 60 working groups for their 8 CUs (three rounds instead of two; measured).
  // ONE slice (a short K: the rank-k update of the kinship back-transformation): the grid is the tile list itself.
  const int id = blockIdx.x, xcd = id & 7, wq = id >> 3;
  const int slice = SUB ? 0 : xcd + 8 * (wq / n_tiles);
  if (slice >= n_slices || (SUB && id >= n_tiles)) return;
  int t = SUB ? id : wq % n_tiles, rp = 0, ct = 0;
  if (symmetric) {  // row panel rp holds the column tiles ct >= first(rp) = (rp * BM) / BN
    for (;; ++rp) {
      const int first = (rp * BM) / BN, cnt = gemm_f64_panel_last(rp, n_col_tiles, halo) - first;
      if (t < cnt) {
        ct = first + t;
        break;
      }
      t -= cnt;
    }
  } else {
    rp = t / n_col_tiles;
    ct = t % n_col_tiles;
  }
  const long long m0 = (long long)rp * BM, n0 = (long long)ct * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const long long k_lo = (long long)slice * kslice;
  long long k_hi = k_lo + kslice;
  if (k_hi > K) k_hi = K;
  double* __restrict__ C = C0 + (long long)slice * c_slice;
  const int Ntot = Nb + Nb2;
  const double* gsrc[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int P = wave + kWaves * q;
    const int r = 8 * P + (lane >> 3), slot = lane & 7;
    const int seg = slot ^ ((r >> 1) & 7);
    const double* base;
    if (r < BM) {
      long long m = m0 + r;
      if (m >= M) m = M - 1;  // (rows beyond the matrix read a valid column; their outputs are never stored)
      if (ring > 0) {
        m += col0;
        if (m >= ring) m -= ring;
      }
      base = A + m * lda;
    } else {
      long long j = n0 + (r - BM);
      if (j >= Ntot) j = Ntot - 1;
      if (j < Nb) {
        if (ring > 0) {
          j += col0;
          if (j >= ring) j -= ring;
        }
        base = B + j * ldb;
      } else {
        base = B2 + (j - Nb) * ldb2;
      }
    }
    gsrc[q] = base + k_lo + seg * 2;  // (16-byte segment = 2 samples)
  }
  auto stage = [&](int buf, long long kc) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int P = wave + kWaves * q;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[q] + kc * KC),
                                       (__attribute__((address_space(3))) void*)(&lds[buf][1024 * P]), 16, 0, 0);
    }
  };
  gd4_t acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = gd4_t{0.0, 0.0, 0.0, 0.0};
  const long long nchunks = (k_hi - k_lo + KC - 1) / KC;
  if (nchunks <= 0) return;
  static_assert(NST == 3, "the pipeline below keeps one chunk in use, one landed, one in flight");
  const int lrow = lane & 15, lg = lane >> 4;
  const double* wsrc = w ? w + k_lo + 2 * lg : nullptr;
  gd2_t fa[2][TM], fb[2][TN];
  auto frags = [&](int buf, int ks, int slot) {  // k-step ks of a chunk: samples 8 ks .. 8 ks + 7, this lane holds 8 ks + 2 lg, + 1
    const char* la = &lds[buf][0];
    const char* lb = &lds[buf][BM * 128];
#pragma unroll
    for (int a = 0; a < TM; ++a)
      fa[slot][a] = *reinterpret_cast<const gd2_t*>(la + rot_lds_off(wm * 16 * TM + a * 16 + lrow, ks * 4 + lg));
#pragma unroll
    for (int b = 0; b < TN; ++b)
      fb[slot][b] = *reinterpret_cast<const gd2_t*>(lb + rot_lds_off(wn * 16 * TN + b * 16 + lrow, ks * 4 + lg));
  };
  // the 64 matrix instructions of one k-step pair; `between` (the LDS reads of the NEXT fragments) is issued after the first
  // eight of them, so that the reads a matrix instruction waits for were always requested 56 matrix instructions earlier —
  // the compiler's s_waitcnt in front of the first matrix instruction is lgkmcnt(0): reads issued just before it would be
  // waited for as well
  auto multiply = [&](int slot, gd2_t wv, auto&& between) {
    if (wsrc) {
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[slot][a] = fa[slot][a] * wv;
    }
    if (SUB) {  // accumulate -(A'DB): the epilogue then ADDS the accumulators to C, straight out of their registers
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[slot][a] = -fa[slot][a];
    }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][a][h], fb[slot][b][h], acc[a][b], 0, 0, 0);
      if (a == 0) {
        __builtin_amdgcn_sched_barrier(0);
        between();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // Pipeline (three LDS stages): while chunk kc is multiplied, chunk kc + 1 has landed (or lands) and chunk kc + 2 is
  // requested.  ONE barrier per chunk, in the MIDDLE of its matrix instructions: behind it every wave's loads of chunk
  // kc + 1 are visible and every wave has left chunk kc - 1, whose buffer is refilled at once.
  stage(0, 0);
  stage(1, nchunks > 1 ? 1 : 0);
  rot_wait_vm_barrier<PPW>();  // chunk 0 is there (its PPW loads were issued first)
  frags(0, 0, 0);
  int cur = 0;
  for (long long kc = 0; kc < nchunks; ++kc) {
    int nxt = cur + 1, nn = cur + 2;
    if (nxt >= NST) nxt -= NST;
    if (nn >= NST) nn -= NST;
    gd2_t wv0 = gd2_t{1.0, 1.0}, wv1 = gd2_t{1.0, 1.0};
    if (wsrc) {
      wv0 = *reinterpret_cast<const gd2_t*>(wsrc + kc * KC);
      wv1 = *reinterpret_cast<const gd2_t*>(wsrc + kc * KC + 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    multiply(0, wv0, [&]() { frags(cur, 1, 1); });
    __builtin_amdgcn_sched_barrier(0);
    rot_wait_vm_barrier<0>();
    const long long nx = (kc + 2 < nchunks) ? kc + 2 : nchunks - 1;
    stage(nn, nx);
    __builtin_amdgcn_sched_barrier(0);
    // (past the last chunk the prefetch reads a buffer that holds a valid older chunk; the values are not used)
    multiply(1, wv1, [&]() { frags(nxt, 0, 0); });
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // element r of a lane's tile: row 4 r + (lane >> 4) of the A side, column lane & 15 of the B side
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const long long j = n0 + wn * 16 * TN + b * 16 + lrow;
    if (j >= Ntot) continue;
    double* cj = C + j * ldc;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      if (SUB) {
        // C -= A'DB (one K slice, the grid is the tile list: every entry belongs to ONE thread).  The accumulators hold the
        // NEGATED product (the A fragments were negated on their way in: rounding is symmetric in sign, the sums are the
        // same numbers with the other sign), so the update is a fire-and-forget fp64 atomic add whose data operand is the
        // accumulator register itself.  The load-subtract-store form had to copy all 256 accumulator registers of a lane into
        // VGPRs for the subtraction and spilled (316 B of scratch at 512 registers).
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long m = m0 + wm * 16 * TM + a * 16 + 4 * r + lg;
          if (m < M) unsafeAtomicAdd(&cj[m], acc[a][b][r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long m = m0 + wm * 16 * TM + a * 16 + 4 * r + lg;
          if (m < M) cj[m] = acc[a][b][r];
        }
      }
    }
  }
}

}  // namespace rvt
