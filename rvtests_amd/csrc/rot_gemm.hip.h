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
// 128 (columns), K chunks of 64 bytes staged through LDS (XOR-swizzled 16-byte segments: conflict-free ds_read_b128 in
// the MFMA operand layout), register prefetch of the next chunk, double-buffered LDS, 8 waves of 64 x 64.  Workgroups
// are mapped so that the 32 workgroups resident on one XCD work on 4 row panels x 8 column tiles at a time: every
// operand chunk is fetched from HBM once per XCD and shared through its L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rvt {

constexpr int kRotBM = 256, kRotBN = 128, kRotKC = 64;  // workgroup tile and K chunk (bytes)
constexpr int kRotPlanesU = 6;                           // digits of U
constexpr int kRotPlanesG = 6;                           // digits of a non-integer column

typedef int i16v_t __attribute__((ext_vector_type(16)));
typedef int i4v_t __attribute__((ext_vector_type(4)));

// position of the 16-byte segment `seg` (0..3) of row `row` inside a [rows][64 B] LDS tile
__device__ __forceinline__ int rot_lds_off(int row, int seg) { return row * 64 + ((seg ^ ((row >> 2) & 3)) << 4); }

// One plane pair.  A: [Mpad rows][ldk] int8 (row m = column m of U), B: [Npad cols][ldk] int8, K = ldk rounded to 64.
// C[m + j * ldc] = (accumulate ? C : 0) + (double)acc * weight * col_scale[j] [* row_scale[m]],  m < M, j < N.
__global__ __launch_bounds__(512, 2) void rot_gemm_i8_kernel(const int8_t* __restrict__ A, const int8_t* __restrict__ B,
                                                             long long ldk, long long kbytes, double* __restrict__ C,
                                                             long long ldc, int M, int N, int n_row_panels,
                                                             int n_col_tiles, const double* __restrict__ col_scale,
                                                             const double* __restrict__ row_scale, double weight,
                                                             int accumulate) {
  __shared__ __attribute__((aligned(16))) char lds[2][(kRotBM + kRotBN) * kRotKC];
  // ---- tile of this workgroup: 32 consecutive workgroups of one XCD = 4 row panels x 8 column tiles ----------------
  const int bid = blockIdx.x, xcd = bid & 7, w = bid >> 3;
  const int n_ctg = (n_col_tiles + 7) / 8;           // column-tile groups
  const int set = w >> 5, within = w & 31;
  const int ctg = set % n_ctg, rpg = set / n_ctg;
  const int rp = (rpg * 8 + xcd) * 4 + (within & 3), ct = ctg * 8 + (within >> 2);
  if (rp >= n_row_panels || ct >= n_col_tiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;           // 4 x 2 waves of 64 x 64
  const long long m0 = (long long)rp * kRotBM, n0 = (long long)ct * kRotBN;
  // ---- global -> register prefetch: (256 + 128) rows x 4 segments = 1536 x 16 B, 3 per thread ------------------------
  const int8_t* gsrc[3];
  int ldst[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int idx = tid + 512 * r;  // 0 .. 1535
    const int row = idx >> 2, seg = idx & 3;
    if (row < kRotBM) {
      gsrc[r] = A + (m0 + row) * ldk + seg * 16;
      ldst[r] = rot_lds_off(row, seg);
    } else {
      gsrc[r] = B + (n0 + (row - kRotBM)) * ldk + seg * 16;
      ldst[r] = kRotBM * 64 + rot_lds_off(row - kRotBM, seg);
    }
  }
  i16v_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0;
  const long long nchunks = kbytes / kRotKC;
  i4v_t pre[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) pre[r] = *reinterpret_cast<const i4v_t*>(gsrc[r]);
#pragma unroll
  for (int r = 0; r < 3; ++r) *reinterpret_cast<i4v_t*>(&lds[0][ldst[r]]) = pre[r];
  __syncthreads();
  const int lrow = lane & 31, lk = lane >> 5;
  for (long long kc = 0; kc < nchunks; ++kc) {
    const int cur = (int)(kc & 1);
    if (kc + 1 < nchunks) {
#pragma unroll
      for (int r = 0; r < 3; ++r) pre[r] = *reinterpret_cast<const i4v_t*>(gsrc[r] + (kc + 1) * kRotKC);
    }
    const char* la = &lds[cur][0];
    const char* lb = &lds[cur][kRotBM * 64];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      i4v_t fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
        fa[a] = *reinterpret_cast<const i4v_t*>(la + rot_lds_off(wm * 64 + a * 32 + lrow, ks * 2 + lk));
#pragma unroll
      for (int b = 0; b < 2; ++b)
        fb[b] = *reinterpret_cast<const i4v_t*>(lb + rot_lds_off(wn * 64 + b * 32 + lrow, ks * 2 + lk));
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (kc + 1 < nchunks) {
#pragma unroll
      for (int r = 0; r < 3; ++r) *reinterpret_cast<i4v_t*>(&lds[cur ^ 1][ldst[r]]) = pre[r];
    }
    __syncthreads();
  }
  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) -----------------
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const long long j = n0 + wn * 64 + b * 32 + (lane & 31);
    if (j >= N) continue;
    const double sc = weight * col_scale[j];
    double* cj = C + j * ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long long m = m0 + wm * 64 + a * 32 + 8 * g + 4 * (lane >> 5);
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

// ---- version 2: LDS-DMA staging, configurable register blocking --------------------------------------------------------
// Same contract and workgroup-to-tile mapping as rot_gemm_i8_kernel.  Differences:
//   * the K chunks go from global memory straight into LDS (global_load_lds_dwordx4: no staging registers and no
//     ds_write_b128 pass — a wide LDS store costs ~13 LDS-path cycles per wave-instruction against 4 for a wide read, and
//     in version 1 the stores took more LDS time than the fragment reads).  The DMA writes lane-linear (base + 16 lane),
//     so the XOR swizzle is applied to the SOURCE address: the lane that fills slot p of row r fetches segment
//     p ^ ((r >> 2) & 3), and the fragment reads use the same involution (rot_lds_off);
//   * WM x WN waves of TM x TN 32x32 tiles each: BM = 32 WM TM rows, BN = 32 WN TN columns.  A k-step of one wave is
//     TM + TN fragment reads for TM TN matrix instructions (version 1: 4 for 4).
template <int WM, int WN, int TM, int TN>
struct RotCfg {
  static constexpr int kWaves = WM * WN, kThreads = 64 * kWaves;
  static constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  static constexpr int kPieces = (BM + BN) / 16;             // 1 KiB pieces (16 rows x 64 B) per K chunk
  static constexpr int kPiecesPerWave = kPieces / kWaves;
  static constexpr int kStageBytes = (BM + BN) * kRotKC;
  static_assert(kPieces % kWaves == 0, "pieces must divide evenly among the waves");
};

template <int WM, int WN, int TM, int TN, int MINB>
__global__ __launch_bounds__(64 * WM * WN, MINB) void rot_gemm_i8_v2_kernel(
    const int8_t* __restrict__ A, const int8_t* __restrict__ B, long long ldk, long long kbytes, double* __restrict__ C,
    long long ldc, int M, int N, int n_row_panels, int n_col_tiles, const double* __restrict__ col_scale,
    const double* __restrict__ row_scale, double weight, int accumulate) {
  using Cfg = RotCfg<WM, WN, TM, TN>;
  __shared__ __attribute__((aligned(1024))) char lds[2][Cfg::kStageBytes];
  const int bid = blockIdx.x, xcd = bid & 7, w = bid >> 3;
  const int n_ctg = (n_col_tiles + 7) / 8;
  const int set = w >> 5, within = w & 31;
  const int ctg = set % n_ctg, rpg = set / n_ctg;
  const int rp = (rpg * 8 + xcd) * 4 + (within & 3), ct = ctg * 8 + (within >> 2);
  if (rp >= n_row_panels || ct >= n_col_tiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const long long m0 = (long long)rp * Cfg::BM, n0 = (long long)ct * Cfg::BN;
  // ---- LDS-DMA sources: piece P = wave + kWaves q covers rows 16 P .. 16 P + 15 of [A rows | B rows] -------------------
  const int8_t* gsrc[Cfg::kPiecesPerWave];
#pragma unroll
  for (int q = 0; q < Cfg::kPiecesPerWave; ++q) {
    const int P = wave + Cfg::kWaves * q;
    const int r = 16 * P + (lane >> 2), slot = lane & 3, seg = slot ^ ((r >> 2) & 3);
    gsrc[q] = (r < Cfg::BM ? A + (m0 + r) * ldk : B + (n0 + (r - Cfg::BM)) * ldk) + seg * 16;
  }
  auto stage = [&](int buf, long long kc) {
#pragma unroll
    for (int q = 0; q < Cfg::kPiecesPerWave; ++q) {
      const int P = wave + Cfg::kWaves * q;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[q] + kc * kRotKC),
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
  const long long nchunks = kbytes / kRotKC;
  stage(0, 0);
  __syncthreads();  // (its fence waits for the DMA: vmcnt(0))
  const int lrow = lane & 31, lk = lane >> 5;
  for (long long kc = 0; kc < nchunks; ++kc) {
    const int cur = (int)(kc & 1);
    if (kc + 1 < nchunks) stage(cur ^ 1, kc + 1);
    const char* la = &lds[cur][0];
    const char* lb = &lds[cur][Cfg::BM * 64];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
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
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();  // chunk kc + 1 has landed (vmcnt(0) in the fence) and every wave is done with buffer `cur`
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
            const double v = (double)acc[a][b][4 * g + e] * (row_scale ? sc * row_scale[m + e] : sc);
            cj[m + e] = accumulate ? cj[m + e] + v : v;
          }
        }
      }
    }
  }
}

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
__global__ void rot_quantize_f32_kernel(const float* __restrict__ src, long long n, long long ncols, long long ld_src,
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

// per-column max |x| of a double matrix (column-major, ld)
__global__ void rot_colmax_kernel(const double* __restrict__ src, long long n, long long ld, double* __restrict__ out) {
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

// double columns -> digit planes.  planes == 1: the entries are integers in [-128, 127], stored as they are (sexp[j]
// must be 0); else entries scaled by 2^sexp[j].
__global__ void rot_quantize_f64_kernel(const double* __restrict__ src, long long n, long long ncols, long long ld_src,
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

}  // namespace rvt
