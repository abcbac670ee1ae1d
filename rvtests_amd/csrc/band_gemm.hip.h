// rvtests_amd — the score-covariance BAND of MetaCovTest's sliding window on a circular device ring.
//
// The reference keeps the genotype vectors of the variants inside the window in RingMemoryPool (base/RingMemoryPool.cpp:31-63:
// allocate / deallocate by index, nothing ever moves) and, when a head leaves the window, prints its covariance with every
// marker still in it (MetaCovTest::printCovariance, src/Model.cpp:942-1004; the window rule src/Model.h:3954-3990).  Here the
// window is a ring of columns in HBM addressed modulo its capacity: logical column j of a call is physical column
// (col0 + j) mod ring.  For hard calls under an unweighted model only the int8 copy of a column (1 byte per genotype, made
// behind its upload) is read: S = G_H' G_W is an exact integer product on the int8 matrix cores.
//
// Only the tiles of the band are enumerated: row panel rp (heads 256 rp .. 256 rp + 255) needs the markers
// 256 rp .. 256 rp + 255 + halo, i.e. ceil((256 + halo) / 256) column tiles that start ON the panel's own diagonal tile
// (a heads x window rectangle computes about twice the band that is printed).  K (the sample index) is split into slices
// that are dealt to the XCDs — XCD x takes the slices x, x + 8, ... and walks the tile list of one slice before the next, so
// the 32 workgroups resident on an XCD read the same sample range of neighbouring columns at the same time and share it
// through that XCD's L2.  Every (slice, tile) writes its own 256 x 256 int32 partial tile, row-major so that the 32 lanes of a
// half-wave store one 128-byte line; band_finish_i32_kernel adds the slices (integers: exact in any order), applies the
// centring / covariate algebra of computeScaledXX (src/Model.h:3997-4005) and writes the band in the layout the adapter
// prints from: band[h * (halo + 1) + t] = value(head h, marker h + t).
//
// The inner loop is the one of rot_gemm.hip.h (global_load_lds_dwordx4 into XOR-swizzled 128-byte rows, two 64 KB stages,
// one barrier per 128 K-bytes, v_mfma_i32_32x32x32_i8, 4 x 2 tiles per wave).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "band_tiles.h"
#include "rot_gemm.hip.h"

namespace rvt {

// RA, RB: the int8 columns, [physical column][ldk] (ldk a multiple of 128, pad rows zero).  part: [slice][tile][256][256] int32.
// grid = 8 * n_tiles * ceil(n_slices / 8) workgroups of 512 threads.
//
// FP4 = true: the columns are stored as 4-bit E2M1 codes, two genotypes per byte (0 -> 0x0, 1 -> 0x2 = 1.0, 2 -> 0x4 = 2.0:
// hard calls are exactly representable in MXFP4), and the tile products run on v_mfma_scale_f32_32x32x64_f8f6f4 with unit
// block scales (E8M0 127): the same 16-byte fragment per lane now carries 32 samples instead of 16, the matrix instruction
// contracts 64 samples at the rate the int8 one contracts 32, and every byte moved (HBM, L2, LDS) carries two genotypes.  The
// accumulation is fp32 and EXACT: all products are in {0, 1, 2, 4}, a slice's partial sum is an integer below 2^24 (the host
// keeps 4 x slice length under it), and fp32 adds integers below 2^24 without rounding in any order.  The partial tiles are
// stored as int32 like the int8 kernel's — everything behind the product is shared.  A stage row of 128 bytes = 256 samples.
typedef float f16v_t __attribute__((ext_vector_type(16)));
typedef int i8v_t __attribute__((ext_vector_type(8)));
template <int WM, int WN, int TM, int TN, int NST, int KC, bool FP4 = false>
__global__ __launch_bounds__(64 * WM * WN, 1) void band_gemm_i8_kernel(const int8_t* __restrict__ RA,
                                                                        const int8_t* __restrict__ RB, long long ldk, int ring,
                                                                        int col0, int H, int W, int halo, long long kbytes0,
                                                                        long long kslice, int n_slices, int n_tiles,
                                                                        int* __restrict__ part) {
  // RA / RB: the column stores the HEAD side and the MARKER side of a tile are read from — the same store for the plain band;
  // the hard-call parts and the other-value masks of mean-imputed columns in any of the four combinations (band_rows.hip.h)
  constexpr int kWaves = WM * WN, BM = 32 * WM * TM, BN = 32 * WN * TN;
  static_assert(BM == kBandBT && BN == kBandBT, "band tiles are 256 x 256");
  static_assert(KC == 128, "a stage holds 128 bytes of K per row");
  constexpr int kPieces = (BM + BN) / 8, PPW = kPieces / kWaves;  // 1 KiB pieces: 8 rows x 128 B
  constexpr int kStage = (BM + BN) * KC;
  static_assert(kPieces % kWaves == 0, "pieces must divide evenly among the waves");
  static_assert(PPW * (NST - 1) < 64, "the ring's loads must fit the vmcnt counter");
  __shared__ __attribute__((aligned(1024))) char lds[NST][kStage];
  const int id = blockIdx.x, xcd = id & 7, wq = id >> 3;
  const int slice = xcd + 8 * (wq / n_tiles);
  if (slice >= n_slices) return;
  const int tile = wq % n_tiles;
  int rp = 0, ct = 0;
  {
    int t = tile;
    for (;; ++rp) {
      const int cnt = band_panel_tiles(rp, W, halo);
      if (t < cnt) {
        ct = rp + t;
        break;
      }
      t -= cnt;
    }
  }
  const long long k_off = (long long)slice * kslice;
  const long long kbytes = (kbytes0 - k_off < kslice) ? kbytes0 - k_off : kslice;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = rp * BM, n0 = ct * BN;
  const int8_t* gsrc[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int P = wave + kWaves * q;
    const int r = 8 * P + (lane >> 3), slot = lane & 7;
    const int seg = slot ^ ((r >> 1) & 7);
    int L = (r < BM) ? m0 + r : n0 + (r - BM);  // logical column of the window
    if (L >= W) L = W - 1;                      // (rows beyond the window read a valid column; their outputs are never used)
    long long phys = (long long)col0 + L;
    if (ring > 0 && phys >= ring) phys -= ring;
    gsrc[q] = (r < BM ? RA : RB) + phys * ldk + k_off + seg * 16;
  }
  auto stage = [&](int buf, long long kc) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int P = wave + kWaves * q;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[q] + kc * KC),
                                       (__attribute__((address_space(3))) void*)(&lds[buf][1024 * P]), 16, 0, 0);
    }
  };
  typename std::conditional<FP4, f16v_t, i16v_t>::type acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0;
  const long long nchunks = kbytes / KC;
  if (nchunks <= 0) return;
#pragma unroll
  for (int t = 0; t < NST - 1; ++t) stage(t, t < nchunks ? t : nchunks - 1);
  const int lrow = lane & 31, lk = lane >> 5;
  int cur = 0;
  constexpr int KS = KC / 32;  // k-steps (32 bytes of K per matrix instruction) per stage
  for (long long kc = 0; kc < nchunks; ++kc) {
    rot_wait_vm_barrier<PPW*(NST - 2)>();
    // the stage that is refilled during this iteration: the buffer every wave left at the barrier above
    const long long nx = (kc + NST - 1 < nchunks) ? kc + NST - 1 : nchunks - 1;
    int nbuf = cur + NST - 1;
    if (nbuf >= NST) nbuf -= NST;
    stage(nbuf, nx);
    const char* la = &lds[cur][0];
    const char* lb = &lds[cur][BM * KC];
    i4v_t fa[2][TM], fb[2][TN];
    auto frags = [&](int ks, int slot) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
        fa[slot][a] = *reinterpret_cast<const i4v_t*>(la + rot_lds_off(wm * 32 * TM + a * 32 + lrow, ks * 2 + lk));
#pragma unroll
      for (int b = 0; b < TN; ++b)
        fb[slot][b] = *reinterpret_cast<const i4v_t*>(lb + rot_lds_off(wn * 32 * TN + b * 32 + lrow, ks * 2 + lk));
    };
    frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks < KS - 1) frags(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the next k-step's reads in front of this k-step's matrix instructions
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          if constexpr (FP4) {
            const i4v_t x = fa[ks & 1][a], y = fb[ks & 1][b];
            const i8v_t xa = {x[0], x[1], x[2], x[3], 0, 0, 0, 0}, yb = {y[0], y[1], y[2], y[3], 0, 0, 0, 0};
            // (cbsz = blgp = 4: both operands E2M1; scales 0x7f = 2^0 in every byte)
            acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, yb, acc[a][b], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          } else {
            acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ks & 1][a], fb[ks & 1][b], acc[a][b], 0, 0, 0);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" ::: "memory");
    if (++cur == NST) cur = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // element 4 g + e of a lane's 32 x 32 tile: row 8 g + 4 (lane >> 5) + e of the head side, column lane & 31 of the marker side
  int* __restrict__ out = part + (((long long)slice * n_tiles + tile) << 16);
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ml = wm * 32 * TM + a * 32 + 8 * g + 4 * (lane >> 5) + e;
#pragma unroll
        for (int b = 0; b < TN; ++b) out[ml * kBandBT + wn * 32 * TN + b * 32 + (lane & 31)] = (int)acc[a][b][4 * g + e];
      }
}
#define band_gemm_i8 (band_gemm_i8_kernel<2, 4, 4, 2, 2, 128>)
#define band_gemm_fp4 (band_gemm_i8_kernel<2, 4, 4, 2, 2, 128, true>)
constexpr int kBandThreads = 512;

}  // namespace rvt
