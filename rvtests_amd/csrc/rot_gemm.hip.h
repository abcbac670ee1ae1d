// rvtests_amd — the kinship rotation  G~ = U' G  (regression/FamSkat.cpp:77-93 multiplies wg with the N x N P0 / Sigma^-1;
// with G~ = U'G everything becomes diagonal-weighted; FastLMM::TransformCentered,
// regression/FastLMM.cpp:611-625) as an EXACT INTEGER matrix product on the int8 matrix cores.
//
// U (the eigenvectors of the kinship, float at the boundary: EigenMatrix = Eigen::MatrixXf,
// regression/EigenMatrix.h:9-12) is stored on the device as fixed-point digits: u = 2^-s * sum_p d_p 128^p with P signed
// base-128 digits d_p in [-64, 63] (P = 6: 40 fractional bits, |error| <= 2^-41 per entry — finer than the float the
// entry came from for every |u| >= 2^-17; all digits of an entry are exact integers, so nothing else is lost).  A batch
// of genotype columns is either small integers (hard calls after flip-to-minor, collapsed burden columns: ONE digit
// plane, exact) or is quantised the same way with a per-column scale.  Then
//       G~[k, j] = 2^-(sU + s_j) * sum_{p, q} 128^(p + q) * ( D_p' E_q )[k, j]
// and every D_p' E_q is an int8 x int8 -> int32 GEMM (v_mfma_i32_32x32x32_i8, ~4.4 POP/s dense on MI355X against
// 78.6 TFLOP/s for the fp64 matrix cores the dgemm it replaces ran on), accumulated exactly; the P (x Q) partial
// products are added into the fp64 result from the least significant plane up.  U takes 6 N^2 bytes (60 GB at
// N = 100 000) instead of 8 N^2 as doubles.
//
// Kernel: TN GEMM, both operands contiguous along the contraction (sample) index.  Workgroup tile 256 (rows of U') x
// 256 (columns), 8 waves as 2 x 4, each wave 4 x 2 tiles of v_mfma_i32_32x32x32_i8 (6 fragment reads per 8 matrix
// instructions).  K is consumed in chunks of 128 bytes — whole cache lines — that go from global memory straight into
// LDS (global_load_lds_dwordx4: no staging registers, no ds_write pass); the DMA writes lane-linear, so the XOR swizzle
// of the 16-byte segments of a row (slot = seg ^ ((row >> 1) & 7): the 16 lanes of a ds_read_b128 group hit 16
// different (row parity, slot) pairs = all 64 banks) is applied to the SOURCE address and again by the fragment reads.
// Two LDS stages (128 KB): iteration kc waits (counted vmcnt) for its own chunk, passes ONE raw barrier — after which
// every wave is done with chunk kc - 1, whose buffer is refilled at once — and computes; the fragments of k-step s + 1
// are requested before the matrix instructions of k-step s.  Workgroups are mapped so that the 32 workgroups resident
// on one XCD work on 4 row panels x 8 column tiles at a time: every operand chunk is fetched from HBM once per XCD and
// shared through its L2.
// Measured ladder at N = 100 000, T = 3840 (tools/rotgemm_bench, POP/s per plane pair): 256 x 128 tile, 2 x 2 tiles per
// wave, register staging + ds_write, 64-byte chunks 1.23; the same with LDS-DMA 1.24 (the stores were not the limit);
// 256 x 256 tile, 4 x 2 per wave 1.79; + 3 stages with counted vmcnt 2.01; + 128-byte chunks and fragment prefetch 2.11.
// Round 4 (tools/rotgemm_bench with -DRVT_ROT_CFG / -DRVT_ROT_DIAG / -DRVT_ROT_SPREAD; the template takes the K bytes of a
// stage, 64 or 128): a ring of five / four / three 32 KB stages of 64 K-bytes — 96-128 KB of loads in flight instead of 64 —
// gives 2.01 / 2.01 / 1.98 POP/s against 2.23 for two 64 KB stages: the prefetch distance is NOT what bounds this kernel
// (twice the barriers cost 10 %).  Timing-only builds: without any refill after the prologue 2.88 POP/s, without the fragment
// reads of k-steps 1-3 2.35, with the refill's pieces spread between the matrix instructions instead of issued at the top of
// the iteration 2.08.  So the matrix instructions with their barrier per 128 K-bytes run at 2.9 of the 3.9 POP/s ceiling by
// themselves, and the LDS-DMA refill costs another 23 % that neither depth nor placement gives back.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rvt {

constexpr int kRotBM = 256, kRotBN = 256, kRotKC = 128;  // workgroup tile and K chunk (bytes)
constexpr int kRotPlanesU = 6;                            // digits of U
constexpr int kRotPlanesG = 6;                            // digits of a non-integer column

typedef int i16v_t __attribute__((ext_vector_type(16)));
typedef int i4v_t __attribute__((ext_vector_type(4)));

// position of the 16-byte segment `seg` (0..7) of row `row` inside a [rows][128 B] LDS tile
__device__ __forceinline__ int rot_lds_off(int row, int seg) { return row * 128 + ((seg ^ ((row >> 1) & 7)) << 4); }

// the same for rows of KC bytes (KC = 128: the function above; KC = 64: four segments per row, slot = seg ^ ((row >> 2) & 3) —
// the 16 lanes of a ds_read_b128 group read rows r = 0..3 mod 4 four times each, with four different (row >> 2) & 3)
template <int KC>
__device__ __forceinline__ int rot_lds_off_t(int row, int seg) {
  if (KC == 128) return rot_lds_off(row, seg);
  return row * 64 + ((seg ^ ((row >> 2) & 3)) << 4);
}

template <int N>
__device__ __forceinline__ void rot_wait_vm_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// One plane pair.  A: [Mpad rows][ldk] int8 (row m = column m of U), B: [Npad cols][ldk] int8, kbytes = K rounded up to 128.
// C[m + j * ldc] = (accumulate ? C : 0) + (double)acc * weight * col_scale[j] [* row_scale[m]],  m < M, j < N.
// grid = (sets * 256, K slices) workgroups, sets = ceil(n_row_panels / 32) * ceil(n_col_tiles / 8); with one slice pass
// kslice = kbytes and c_slice = 0.
template <int WM, int WN, int TM, int TN, int NST, int MINB, int KC>
__global__ __launch_bounds__(64 * WM * WN, MINB) void rot_gemm_i8_kernel_t(
    const int8_t* __restrict__ A0, const int8_t* __restrict__ B0, long long ldk, long long kbytes0, double* __restrict__ C0,
    long long ldc, int M, int N, int n_row_panels, int n_col_tiles, const double* __restrict__ col_scale,
    const double* __restrict__ row_scale, double weight, int accumulate, long long kslice, long long c_slice) {
  constexpr int kWaves = WM * WN, BM = 32 * WM * TM, BN = 32 * WN * TN;
  // split K: slice blockIdx.y covers K bytes [y kslice, (y + 1) kslice) and writes its own result at C0 + y c_slice
  const long long k_off = (long long)blockIdx.y * kslice;
  const int8_t* __restrict__ A = A0 + k_off;
  const int8_t* __restrict__ B = B0 + k_off;
  double* __restrict__ C = C0 + (long long)blockIdx.y * c_slice;
  const long long kbytes = (kbytes0 - k_off < kslice) ? kbytes0 - k_off : kslice;
  static_assert(KC == 64 || KC == 128, "a stage holds 64 or 128 bytes of K per row");
  constexpr int kRowsPerPiece = 1024 / KC, kSegs = KC / 16;                   // 1 KiB pieces: 8 rows x 128 B or 16 rows x 64 B
  constexpr int kPieces = (BM + BN) / kRowsPerPiece, PPW = kPieces / kWaves;
  constexpr int kStage = (BM + BN) * KC;
  static_assert(kPieces % kWaves == 0, "pieces must divide evenly among the waves");
  static_assert(PPW * (NST - 1) < 64, "the ring's loads must fit the vmcnt counter");
  __shared__ __attribute__((aligned(1024))) char lds[NST][kStage];
  // (K slices rotate the XCD that takes a given tile: a product with few tiles still spreads over the chip)
  const int bid = blockIdx.x, xcd = (bid + blockIdx.y) & 7, w = bid >> 3;
  const int n_ctg = (n_col_tiles + 7) / 8;
  const int set = w >> 5, within = w & 31;
  const int ctg = set % n_ctg, rpg = set / n_ctg;
  const int rp = (rpg * 8 + xcd) * 4 + (within & 3), ct = ctg * 8 + (within >> 2);
  if (rp >= n_row_panels || ct >= n_col_tiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const long long m0 = (long long)rp * BM, n0 = (long long)ct * BN;
  const int8_t* gsrc[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int P = wave + kWaves * q;
    const int r = kRowsPerPiece * P + lane / kSegs, slot = lane % kSegs;
    const int seg = (KC == 128) ? (slot ^ ((r >> 1) & 7)) : (slot ^ ((r >> 2) & 3));
    gsrc[q] = (r < BM ? A + (m0 + r) * ldk : B + (n0 + (r - BM)) * ldk) + seg * 16;
  }
  auto stage = [&](int buf, long long kc) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int P = wave + kWaves * q;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[q] + kc * KC),
                                       (__attribute__((address_space(3))) void*)(&lds[buf][1024 * P]), 16, 0, 0);
    }
  };
  i16v_t acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0;
  const long long nchunks = kbytes / KC;
#pragma unroll
  for (int t = 0; t < NST - 1; ++t) stage(t, t < nchunks ? t : nchunks - 1);
  const int lrow = lane & 31, lk = lane >> 5;
  int cur = 0;
  constexpr int KS = KC / 32;  // k-steps (32 bytes of K per matrix instruction) per stage
  auto stage_piece = [&](int buf, long long kc, int q) {
    const int P = wave + kWaves * q;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[q] + kc * KC),
                                     (__attribute__((address_space(3))) void*)(&lds[buf][1024 * P]), 16, 0, 0);
  };
  for (long long kc = 0; kc < nchunks; ++kc) {
    rot_wait_vm_barrier<PPW*(NST - 2)>();
    // the stage that is refilled during this iteration: the buffer every wave left at the barrier above
    const long long nx = (kc + NST - 1 < nchunks) ? kc + NST - 1 : nchunks - 1;
    int nbuf = cur + NST - 1;
    if (nbuf >= NST) nbuf -= NST;
#if !defined(RVT_ROT_SPREAD)
#if !defined(RVT_ROT_DIAG) || RVT_ROT_DIAG != 1   // (diagnostic builds of tools/rotgemm_bench: 1 = no loads after the prologue,
    stage(nbuf, nx);                                //  2 = no fragment reads after the first k-step — wrong results, timing only)
#endif
#endif
    const char* la = &lds[cur][0];
    const char* lb = &lds[cur][BM * KC];
    i4v_t fa[2][TM], fb[2][TN];
    auto frags = [&](int ks, int slot) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
        fa[slot][a] = *reinterpret_cast<const i4v_t*>(la + rot_lds_off_t<KC>(wm * 32 * TM + a * 32 + lrow, ks * 2 + lk));
#pragma unroll
      for (int b = 0; b < TN; ++b)
        fb[slot][b] = *reinterpret_cast<const i4v_t*>(lb + rot_lds_off_t<KC>(wn * 32 * TN + b * 32 + lrow, ks * 2 + lk));
    };
    frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#if !defined(RVT_ROT_DIAG) || RVT_ROT_DIAG != 2
      if (ks < KS - 1) frags(ks + 1, (ks + 1) & 1);
#else
      if (ks < KS - 1) { for (int a = 0; a < TM; ++a) fa[(ks + 1) & 1][a] = fa[ks & 1][a]; for (int b = 0; b < TN; ++b) fb[(ks + 1) & 1][b] = fb[ks & 1][b]; }
#endif
      __builtin_amdgcn_sched_barrier(0);  // keep the next k-step's reads in front of this k-step's matrix instructions
#if defined(RVT_ROT_SPREAD)
      // the refill's pieces spread over the k-steps, each between matrix instructions (a burst of pieces at the top of the
      // iteration stalls the wave's issue for ~150 cycles per piece)
      constexpr int PPK = (PPW + KS - 1) / KS;
      int qn = ks * PPK;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ks & 1][a], fb[ks & 1][b], acc[a][b], 0, 0, 0);
          if (((a * TN + b) & 1) == 1 && qn < (ks + 1) * PPK && qn < PPW) stage_piece(nbuf, nx, qn++);
        }
#else
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ks & 1][a], fb[ks & 1][b], acc[a][b], 0, 0, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" ::: "memory");
    if (++cur == NST) cur = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const long long j = n0 + wn * 32 * TN + b * 32 + (lane & 31);
    if (j >= N) continue;
    const double sc = weight * col_scale[j];
    double* cj = C + j * ldc;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long long m = m0 + wm * 32 * TM + a * 32 + 8 * g + 4 * (lane >> 5);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (m + e < M) {
            const double v = (double)acc[a][b][4 * g + e] * (row_scale ? sc * row_scale[m + e] : sc);
            cj[m + e] = accumulate ? cj[m + e] + v : v;
          }
        }
      }
    }
  }
}

// Structured A — the eigenvectors of a block-diagonal kinship after rvt_set_kinship has clustered them by support:
// a_krange[row panel] = the K chunks [x, y) that hold every non-zero of those 256 rows; everything else contributes exact
// zeros and is skipped.  A row panel meets only a few K chunks, so one plane pair is a handful of matrix
// instructions per tile and the fp64 read-modify-write of C (8 bytes per output per plane pair) would dominate.  This
// variant walks ALL planes of A inside the kernel — int32 tile per plane, folded into fp64 registers with the plane's
// weight, least significant plane first: the same sums in the same order as one launch per plane — and writes C once.
// One LDS stage (the K loop is too short to pipeline), 8 waves as WM x WN, TM x TN tiles per wave.
// C[m + j ldc] = (accumulate ? C : 0) + sum_p (double)acc_p 128^p * weight * col_scale[j]
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * WM * WN, 1) void rot_gemm_i8_short_kernel(
    const int8_t* __restrict__ A0, long long a_plane_stride, int planes_a, const int8_t* __restrict__ B0, long long ldk,
    long long kbytes0, double* __restrict__ C, long long ldc, int M, int N, int n_row_panels, int n_col_tiles,
    const double* __restrict__ col_scale, double weight, int accumulate, const int2* __restrict__ a_krange) {
  constexpr int kWaves = WM * WN, BM = 32 * WM * TM, BN = 32 * WN * TN;
  constexpr int kPieces = (BM + BN) / 8, PPW = kPieces / kWaves;
  static_assert(kPieces % kWaves == 0 && BM == kRotBM, "row panels are kRotBM rows");
  __shared__ __attribute__((aligned(1024))) char lds[(BM + BN) * kRotKC];
  const int bid = blockIdx.x, xcd = bid & 7, w = bid >> 3;
  const int n_ctg = (n_col_tiles + 7) / 8;
  const int set = w >> 5, within = w & 31;
  const int ctg = set % n_ctg, rpg = set / n_ctg;
  const int rp = (rpg * 8 + xcd) * 4 + (within & 3), ct = ctg * 8 + (within >> 2);
  if (rp >= n_row_panels || ct >= n_col_tiles) return;
  const int2 kr = a_krange[rp];
  long long k_lo = (long long)kr.x * kRotKC, k_hi = (long long)kr.y * kRotKC;
  if (k_hi > kbytes0) k_hi = kbytes0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const long long m0 = (long long)rp * BM, n0 = (long long)ct * BN;
  const int8_t* gsrc[PPW];
  bool is_a[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int P = wave + kWaves * q;
    const int r = 8 * P + (lane >> 3), slot = lane & 7, seg = slot ^ ((r >> 1) & 7);
    is_a[q] = r < BM;
    gsrc[q] = (r < BM ? A0 + (m0 + r) * ldk : B0 + (n0 + (r - BM)) * ldk) + seg * 16;
  }
  double accd[TM][TN][16];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) accd[a][b][e] = 0.0;
  const int lrow = lane & 31, lk = lane >> 5;
  const char* la = &lds[0];
  const char* lb = &lds[BM * 128];
  double pw = 1.0;  // 128^p
  for (int p = 0; p < planes_a; ++p, pw *= 128.0) {
    i16v_t acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0;
    for (long long kb = k_lo; kb < k_hi; kb += kRotKC) {
#pragma unroll
      for (int q = 0; q < PPW; ++q) {
        const int P = wave + kWaves * q;
        const int8_t* src = gsrc[q] + kb + (is_a[q] ? (long long)p * a_plane_stride : 0ll);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(&lds[1024 * P]), 16, 0, 0);
      }
      rot_wait_vm_barrier<0>();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        i4v_t fa[TM], fb[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
          fa[a] = *reinterpret_cast<const i4v_t*>(la + rot_lds_off(wm * 32 * TM + a * 32 + lrow, ks * 2 + lk));
#pragma unroll
        for (int b = 0; b < TN; ++b)
          fb[b] = *reinterpret_cast<const i4v_t*>(lb + rot_lds_off(wn * 32 * TN + b * 32 + lrow, ks * 2 + lk));
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has read this chunk before the next lands
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) accd[a][b][e] += (double)acc[a][b][e] * pw;  // (exact product, one rounding per sum)
  }
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const long long j = n0 + wn * 32 * TN + b * 32 + (lane & 31);
    if (j >= N) continue;
    const double sc = weight * col_scale[j];
    double* cj = C + j * ldc;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long long m = m0 + wm * 32 * TM + a * 32 + 8 * g + 4 * (lane >> 5);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (m + e < M) {
            const double v = accd[a][b][4 * g + e] * sc;
            cj[m + e] = accumulate ? cj[m + e] + v : v;
          }
        }
      }
    }
  }
}
constexpr int kRotShortBN = 128;  // column tile of the short-K kernel (<4, 2, 2, 2>: 256 x 128 per workgroup)
#define rot_gemm_i8_short (rot_gemm_i8_short_kernel<4, 2, 2, 2>)

// the shipped configuration
#ifndef RVT_ROT_CFG
#define RVT_ROT_CFG 2, 4, 4, 2, 2, 1, 128  // (2 x 4 waves, 4 x 2 tiles per wave, two 64 KB stages of 128 K-bytes)
#define RVT_ROT_THREADS 512
#endif
#define rot_gemm_i8_kernel (rot_gemm_i8_kernel_t<RVT_ROT_CFG>)
constexpr int kRotThreads = RVT_ROT_THREADS;

// C[m + j ldc] = (accumulate ? C : 0) + sum over slices of part[s * stride + m + j ldc], m < M, j < N (fixed order:
// reproducible; rows M .. ldc-1 of C are not touched)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void rot_reduce_slices_kernel(const double* __restrict__ part, long long ldc, long long M, long long N,
                                         long long stride, int slices, double* __restrict__ C, int accumulate) {
  const long long total = M * N;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long i = (t / M) * ldc + t % M;
    double s = accumulate ? C[i] : 0.0;
    for (int k = 0; k < slices; ++k) s += part[(long long)k * stride + i];
    C[i] = s;
  }
}
#endif  // RVT_K_FAM

// ---- digits ------------------------------------------------------------------------------------------------------------
// q = sum_p d_p 128^p, d_p in [-64, 63]
__device__ __forceinline__ void rot_digits(long long q, int planes, signed char* d) {
  for (int p = 0; p < planes; ++p) {
    const long long r = ((q + 64) & 127) - 64;  // q mod 128 in [-64, 63] (two's complement & works for negative q)
    d[p] = (signed char)r;
    q = (q - r) >> 7;
  }
}

// float matrix (column-major, n x ncols, leading dimension lds_src) -> digit planes [plane][col][ldk]; entries scaled
// by 2^sexp.  flag[0] is set when an entry does not fit (|u| * 2^sexp >= 2^(7 planes - 2)).
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void rot_quantize_f32_kernel(const float* __restrict__ src, long long n, long long ncols, long long ld_src,
                                        int sexp, int planes, signed char* __restrict__ dst, long long ldk,
                                        long long plane_stride, long long col0, int* __restrict__ flag) {
  const double scale = ldexp(1.0, sexp), lim = ldexp(1.0, 7 * planes - 2);
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n * ncols;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long j = idx / n, i = idx % n;
    const double v = (double)src[j * ld_src + i] * scale;
    if (!(fabs(v) < lim)) {
      *flag = 1;
      continue;
    }
    signed char d[8];
    rot_digits(llrint(v), planes, d);
    for (int p = 0; p < planes; ++p) dst[p * plane_stride + (col0 + j) * ldk + i] = d[p];
  }
}
#endif  // RVT_K_FAM

// first / last row with a non-zero entry per column of a float matrix (lo = n, hi = -1 for an all-zero column)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void rot_span_kernel(const float* __restrict__ src, long long n, long long ld_src, int* __restrict__ lo,
                                int* __restrict__ hi) {
  __shared__ int slo[256], shi[256];
  const float* s = src + (long long)blockIdx.x * ld_src;
  int l = (int)n, h = -1;
  for (long long i = threadIdx.x; i < n; i += blockDim.x)
    if (s[i] != 0.0f) {
      l = l < (int)i ? l : (int)i;
      h = h > (int)i ? h : (int)i;
    }
  slo[threadIdx.x] = l;
  shi[threadIdx.x] = h;
  __syncthreads();
  for (int w = blockDim.x / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      slo[threadIdx.x] = min(slo[threadIdx.x], slo[threadIdx.x + w]);
      shi[threadIdx.x] = max(shi[threadIdx.x], shi[threadIdx.x + w]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    lo[blockIdx.x] = slo[0];
    hi[blockIdx.x] = shi[0];
  }
}
#endif  // RVT_K_FAM

// dst row r = src row order[r] of a [rows][ldk] byte matrix (ldk a multiple of 16); grid = rows
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void rot_gather_rows_kernel(const signed char* __restrict__ src, const int* __restrict__ order, long long ldk,
                                       signed char* __restrict__ dst) {
  const uint4* s = reinterpret_cast<const uint4*>(src + (long long)order[blockIdx.x] * ldk);
  uint4* d = reinterpret_cast<uint4*>(dst + (long long)blockIdx.x * ldk);
  for (long long i = threadIdx.x; i < ldk / 16; i += blockDim.x) d[i] = s[i];
}
#endif  // RVT_K_FAM

// ---- sparse eigenvectors (families interleaved in the sample order) --------------------------------------------------------
// non-zeros per column of a float matrix; grid = columns
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void rot_nnz_count_kernel(const float* __restrict__ src, long long n, long long ld_src, int* __restrict__ count) {
  __shared__ int red[256];
  const float* s = src + (long long)blockIdx.x * ld_src;
  int c = 0;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) c += (s[i] != 0.0f) ? 1 : 0;
  red[threadIdx.x] = c;
  __syncthreads();
  for (int w = blockDim.x / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) count[blockIdx.x] = red[0];
}
#endif  // RVT_K_FAM
// the non-zeros of column blockIdx.x, in ascending row order, to rows / vals at offset colptr[blockIdx.x]; 256 threads
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ __launch_bounds__(256) void rot_nnz_fill_kernel(const float* __restrict__ src, long long n, long long ld_src,
                                                           const long long* __restrict__ colptr, int* __restrict__ rows,
                                                           double* __restrict__ vals) {
  __shared__ int wsum[4];
  __shared__ long long base;
  const float* s = src + (long long)blockIdx.x * ld_src;
  if (threadIdx.x == 0) base = colptr[blockIdx.x];
  __syncthreads();
  for (long long i0 = 0; i0 < n; i0 += 256) {
    const long long i = i0 + threadIdx.x;
    const float v = i < n ? s[i] : 0.0f;
    const bool nz = v != 0.0f;
    const unsigned long long m = __ballot(nz);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int before = __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (nz) {
      rows[base + before] = (int)i;
      vals[base + before] = (double)v;
    }
    __syncthreads();
    if (threadIdx.x == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}
#endif  // RVT_K_FAM
// out[k + t ld_dst] = sum over the non-zeros e of column k of U: vals[e] * G[rows[e] + t ld_src]   (U' G for sparse U);
// grid (ceil(n / 256), columns of G)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ __launch_bounds__(256) void rot_sparse_kernel(const long long* __restrict__ colptr, const int* __restrict__ rows,
                                                         const double* __restrict__ vals, long long n,
                                                         const double* __restrict__ G, long long ld_src,
                                                         double* __restrict__ out, long long ld_dst) {
  const long long k = blockIdx.x * 256ll + threadIdx.x;
  if (k >= n) return;
  const double* g = G + (long long)blockIdx.y * ld_src;
  double s = 0.0;
  for (long long e = colptr[k]; e < colptr[k + 1]; ++e) s = fma(vals[e], g[rows[e]], s);
  out[k + (long long)blockIdx.y * ld_dst] = s;
}
#endif  // RVT_K_FAM

// per-column max |x| of a double matrix (column-major, ld)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void rot_colmax_kernel(const double* __restrict__ src, long long n, long long ld, double* __restrict__ out) {
  __shared__ double red[256];
  const double* s = src + (long long)blockIdx.x * ld;
  double m = 0.0;
  bool small_int = true;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    const double v = s[i];
    m = fmax(m, fabs(v));
    small_int = small_int && (v == rint(v));
  }
  red[threadIdx.x] = small_int ? m : -m - 1.0;  // negative: some entry is not an integer
  __syncthreads();
  for (int w = blockDim.x / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      const double a = red[threadIdx.x], b = red[threadIdx.x + w];
      const double ma = a < 0 ? -a - 1.0 : a, mb = b < 0 ? -b - 1.0 : b;
      const double mm = fmax(ma, mb);
      red[threadIdx.x] = (a < 0 || b < 0) ? -mm - 1.0 : mm;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];  // >= 0: integer column with this max; < 0: -(max) - 1, not integer
}
#endif  // RVT_K_FAM

// double columns -> digit planes.  planes == 1: the entries are integers in [-128, 127], stored as they are (sexp[j]
// must be 0); else entries scaled by 2^sexp[j].
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void rot_quantize_f64_kernel(const double* __restrict__ src, long long n, long long ncols, long long ld_src,
                                        const int* __restrict__ sexp, int planes, signed char* __restrict__ dst,
                                        long long ldk, long long plane_stride) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n * ncols;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long j = idx / n, i = idx % n;
    const double x = src[j * ld_src + i];
    if (planes == 1) {
      dst[j * ldk + i] = (signed char)(int)x;
    } else {
      signed char d[8];
      rot_digits(llrint(ldexp(x, sexp[j])), planes, d);
      for (int p = 0; p < planes; ++p) dst[p * plane_stride + j * ldk + i] = d[p];
    }
  }
}
#endif  // RVT_K_FAM

}  // namespace rvt
