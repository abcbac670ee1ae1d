// rvtests_amd — related-sample path (FastLMM null model + FamSKAT): the small kernels around the shared
// sufficient-statistics machinery.
//
// FamSkat (regression/FamSkat.cpp:34-138) works with N x N matrices Sigma = sigma2 U (S + delta) U', Sigma^-1 and
// P0 = Sigma - X (X' Sigma^-1 X)^-1 X'.  With G~ = U'G (one GEMM per batch of genes) everything it needs is a
// diagonal-weighted contraction over the rotated samples,
//     wg P0 wg' = W (G~' V G~  -  (G'X) C^-1 (G'X)') W,   V = sigma2 (S + delta),  G'X = G~'(U'X),  C = X' Sigma^-1 X
//     wg Sigma^-1 r = W G~' V^-1 U'(y - X beta)
//     FastGetAF_j  = 0.5 * sum_i G~_ij u1_i / |S_i| / (u1'|S|^-1 u1),   u1 = U'1        (FastLMM.cpp:402-443)
// i.e. exactly R = G~' D [G~ | X_in | rr_in] of gene_suffstat_mfma in its weighted mode with
//     D = V,  X_in = [ V^-1 U'X | V^-1 u1/|S| ],  rr_in = V^-2 U'(y - X beta).
// N x N objects never exist; U is read once per batch by the rotation GEMM.
#pragma once
#include <hip/hip_runtime.h>
#include "rvt_gene.h"

namespace rvt {

#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void cvt_f32_f64_kernel(const float* __restrict__ in, double* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = (double)in[i];
}
#endif  // RVT_K_FAM

// out[k] = sum_i A[i + k*lda], one workgroup per column, fixed reduction order
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ __launch_bounds__(256) void column_sums_kernel(const double* __restrict__ A, long long n, long long lda,
                                                          double* __restrict__ out) {
  __shared__ double red[256];
  const double* col = A + (long long)blockIdx.x * lda;
  double s = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) s += col[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
#endif  // RVT_K_FAM

// FastLMM getBetaSigma2 / getLogLikelihood (regression/FastLMM.cpp:297-346) need, for one delta,
//   A = ux' D ux, b = ux' D uy, yy = uy' D uy (D = 1/|lambda + delta|) and sum log|lambda + delta|.
// uxy: N x (d+1) column-major, columns 0..d-1 = U'X, column d = U'y.  Each workgroup writes one partial record
// of kLmmRec doubles: A (d x d row-major), b (d), yy, slog.
constexpr int kLmmBlocks = 256;
__host__ __device__ constexpr int lmm_rec_len(int d) { return d * d + d + 2; }

#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ __launch_bounds__(256) void lmm_sums_kernel(const double* __restrict__ uxy, const double* __restrict__ lam,
                                                       long long N, int d, double delta, int take_abs,
                                                       double* __restrict__ partial) {
  extern __shared__ double sm[];  // 256 doubles
  const int rec = lmm_rec_len(d);
  double* out = partial + (long long)blockIdx.x * rec;
  // one quantity at a time keeps the register footprint independent of d (this kernel runs ~120 times per fit)
  for (int q = 0; q < rec; ++q) {
    int a = 0, b = 0, kind;
    if (q < d * d) {
      kind = 0;
      a = q / d;
      b = q % d;
    } else if (q < d * d + d) {
      kind = 1;
      a = q - d * d;
    } else
      kind = (q == d * d + d) ? 2 : 3;
    double s = 0.0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < N; i += 256LL * gridDim.x) {
      const double t = lam[i] + delta;
      const double w = 1.0 / (take_abs ? fabs(t) : t);
      double v;
      if (kind == 0)
        v = uxy[i + a * N] * w * uxy[i + b * N];
      else if (kind == 1)
        v = uxy[i + a * N] * w * uxy[i + (long long)d * N];
      else if (kind == 2)
        v = uxy[i + (long long)d * N] * w * uxy[i + (long long)d * N];
      else
        v = log(fabs(t));
      s += v;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[q] = sm[0];
    __syncthreads();
  }
}
#endif  // RVT_K_FAM

// ---- genotype consolidation on the device (the "next" row of the boundary: raw / packed genotypes) -------------------
// Reproduces, for every column of raw genotypes (missing = negative):
//   GenotypeCounter::add / getAF   (src/GenotypeCounter.h:14-51): AF = 0.5 * sum{0 <= g <= 2} g / nSample, g < 0 or g > 2
//                                  count as missing, nSample counts every sample
//   DataConsolidator::imputeGenotypeToMean (src/DataConsolidator.cpp:217-245): only when the counter saw a missing
//                                  value; `int ac` accumulates every g >= 0 with truncation after each addition
//                                  (quirk #5), an = 2 * #{g >= 0}; g < 0 <- 2 * ac / an
// Three launches, all columns at once and the sample axis cut into kConsolChunk-sample pieces (a 500 000 x 50 gene is
// 3 100 workgroups instead of 50): partial counts per (column, piece) -> per-column AF and fill value (partials summed
// in piece order) -> imputed fp64 columns.  For integer-valued columns (hard calls) the truncating accumulation is an
// exact integer sum; a column that mixes fractional dosages with missing values replays the reference's sequential
// accumulation on one lane.  SRC is double (in place), int8 (packed hard calls) or bed2_t (PLINK 2-bit codes); the
// packed forms are expanded to fp64.
constexpr int kConsolChunk = 8192;  // samples per workgroup
// PLINK .bed storage, SNP-major (libVcf/PlinkInputFile.cpp:24-47, codes libVcf/PlinkInputFile.h:206-209): 4 samples per
// byte, sample p in bits 2(p & 3) .. 2(p & 3)+1 of byte p >> 2; 00 -> 0, 10 -> 1, 11 -> 2, 01 -> missing (-9).  Every
// variant starts on a byte boundary (its row is ceil(N/4) bytes).
struct bed2_t {
  unsigned char b;
};
template <typename SRC>
__device__ __forceinline__ double load_genotype(const SRC* col, long long i) {
  return (double)col[i];
}
template <>
__device__ __forceinline__ double load_genotype<bed2_t>(const bed2_t* col, long long i) {
  const unsigned code = (col[i >> 2].b >> ((i & 3) << 1)) & 3u;
  return code == 0u ? 0.0 : (code == 2u ? 1.0 : (code == 3u ? 2.0 : -9.0));
}
struct ConsolPart {
  double sumAC, ac;
  long long nonneg;
  int flags, pad;  // bit 0: counter saw a missing value; bit 1: a value < 0 exists; bit 2: a fractional value >= 0
                   // (pad: 2-bit rows only — the number of 2s of the part, for the header of the packed-row kernel)
};

template <typename SRC>
__global__ __launch_bounds__(256) void consolidate_count_kernel(const SRC* __restrict__ src, long long src_ld,
                                                                long long N, ConsolPart* __restrict__ parts) {
  __shared__ double s_sum[256], s_ac[256];
  __shared__ long long s_cnt[256];
  __shared__ int s_flag[256];
  const SRC* col = src + (long long)blockIdx.y * src_ld;
  const long long i0 = (long long)blockIdx.x * kConsolChunk;
  const long long i1 = (i0 + kConsolChunk < N) ? i0 + kConsolChunk : N;
  double sumAC = 0.0, ac = 0.0;
  long long nonneg = 0;
  int flags = 0;
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
    const double g = load_genotype(col, i);
    if (g < 0.0) {
      flags |= 3;
    } else {
      if (g <= 2.0)
        sumAC += g;
      else
        flags |= 1;
      ac += g;
      ++nonneg;
      if (g != floor(g)) flags |= 4;
    }
  }
  s_sum[threadIdx.x] = sumAC;
  s_ac[threadIdx.x] = ac;
  s_cnt[threadIdx.x] = nonneg;
  s_flag[threadIdx.x] = flags;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
      s_ac[threadIdx.x] += s_ac[threadIdx.x + off];
      s_cnt[threadIdx.x] += s_cnt[threadIdx.x + off];
      s_flag[threadIdx.x] |= s_flag[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    parts[(long long)blockIdx.y * gridDim.x + blockIdx.x] = ConsolPart{s_sum[0], s_ac[0], s_cnt[0], s_flag[0], 0};
}

// PLINK 2-bit rows: one thread per BYTE (four samples), the three counts by population counts of bit masks — a quarter of
// the iterations of the generic kernel and no floating point (45 -> ~10 microseconds for a 500 000 x 50 gene: the count
// sits between the host copy of a gene and its expansion).  Hard calls: the truncating accumulation is an integer sum.
// (a full specialisation is an ordinary function: it lives in the ONE translation unit that launches it, rvt_stream.hip)
#ifdef RVT_STREAM_UNIT
template <>
__global__ __launch_bounds__(256) void consolidate_count_kernel<bed2_t>(const bed2_t* __restrict__ src, long long src_ld,
                                                                        long long N, ConsolPart* __restrict__ parts) {
  __shared__ unsigned s_n1[256], s_n2[256], s_nm[256];
  const unsigned char* col = reinterpret_cast<const unsigned char*>(src) + (long long)blockIdx.y * src_ld;
  const long long i0 = (long long)blockIdx.x * kConsolChunk;  // (a multiple of 4)
  const long long i1 = (i0 + kConsolChunk < N) ? i0 + kConsolChunk : N;
  unsigned n1 = 0, n2 = 0, nm = 0;
  for (long long b = (i0 >> 2) + threadIdx.x; 4 * b < i1; b += 256) {
    unsigned v = col[b];
    const long long left = i1 - 4 * b;  // samples of this byte that exist (the last byte of a row may carry padding)
    if (left < 4) v &= (1u << (2 * left)) - 1u;
    const unsigned lo = v & 0x55u, hi = (v >> 1) & 0x55u;
    n1 += __popc(hi & ~lo);  // 10 -> 1
    n2 += __popc(hi & lo);   // 11 -> 2
    nm += __popc(lo & ~hi);  // 01 -> missing
  }
  s_n1[threadIdx.x] = n1;
  s_n2[threadIdx.x] = n2;
  s_nm[threadIdx.x] = nm;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      s_n1[threadIdx.x] += s_n1[threadIdx.x + off];
      s_n2[threadIdx.x] += s_n2[threadIdx.x + off];
      s_nm[threadIdx.x] += s_nm[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double ac = (double)s_n1[0] + 2.0 * (double)s_n2[0];
    parts[(long long)blockIdx.y * gridDim.x + blockIdx.x] =
        ConsolPart{ac, ac, (long long)(i1 - i0) - (long long)s_nm[0], s_nm[0] ? 3 : 0, (int)s_n2[0]};  // pad = #(g = 2)
  }
}
// The same count for rows that lie elsewhere on the device in the FILE's layout (rvt_submit_gene_bed_dev: a resident .bed
// matrix, rows ceil(N/4) bytes apart, any alignment): every byte read is also written into the gene's own block, whose rows
// are padded to 16 bytes for the packed-row kernel — the copy costs no pass and no call of its own.
__global__ __launch_bounds__(256) void bed_count_copy_kernel(const unsigned char* __restrict__ src, long long src_ld, long long N,
                                                             ConsolPart* __restrict__ parts, unsigned char* __restrict__ dst,
                                                             long long dst_ld) {
  __shared__ unsigned s_n1[256], s_n2[256], s_nm[256];
  const unsigned char* col = src + (long long)blockIdx.y * src_ld;
  unsigned char* out = dst + (long long)blockIdx.y * dst_ld;
  const long long i0 = (long long)blockIdx.x * kConsolChunk;
  const long long i1 = (i0 + kConsolChunk < N) ? i0 + kConsolChunk : N;
  unsigned n1 = 0, n2 = 0, nm = 0;
  for (long long b = (i0 >> 2) + threadIdx.x; 4 * b < i1; b += 256) {
    unsigned v = col[b];
    const long long left = i1 - 4 * b;
    if (left < 4) v &= (1u << (2 * left)) - 1u;  // (padding bits of a row's last byte: stored as zeros, as the staged copy's pads are)
    out[b] = (unsigned char)v;
    const unsigned lo = v & 0x55u, hi = (v >> 1) & 0x55u;
    n1 += __popc(hi & ~lo);
    n2 += __popc(hi & lo);
    nm += __popc(lo & ~hi);
  }
  s_n1[threadIdx.x] = n1;
  s_n2[threadIdx.x] = n2;
  s_nm[threadIdx.x] = nm;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      s_n1[threadIdx.x] += s_n1[threadIdx.x + off];
      s_n2[threadIdx.x] += s_n2[threadIdx.x + off];
      s_nm[threadIdx.x] += s_nm[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double ac = (double)s_n1[0] + 2.0 * (double)s_n2[0];
    parts[(long long)blockIdx.y * gridDim.x + blockIdx.x] =
        ConsolPart{ac, ac, (long long)(i1 - i0) - (long long)s_nm[0], s_nm[0] ? 3 : 0, (int)s_n2[0]};
  }
}
#endif

// one wave per column: AF and the imputation value
template <typename SRC>
__global__ __launch_bounds__(64) void consolidate_fill_kernel(const SRC* __restrict__ src, long long src_ld,
                                                              long long N, int nparts,
                                                              const ConsolPart* __restrict__ parts,
                                                              double* __restrict__ af_out, double* __restrict__ fill_out) {
  if (threadIdx.x != 0) return;
  const ConsolPart* p = parts + (long long)blockIdx.x * nparts;
  double sumAC = 0.0, ac = 0.0;
  long long nonneg = 0;
  int flags = 0;
  for (int k = 0; k < nparts; ++k) {
    sumAC += p[k].sumAC;
    ac += p[k].ac;
    nonneg += p[k].nonneg;
    flags |= p[k].flags;
  }
  af_out[blockIdx.x] = N ? 0.5 * sumAC / (double)N : -1.0;
  double fill = 0.0;
  if ((flags & 1) && (flags & 2)) {
    const long long an = 2 * nonneg;
    int aci;
    if (flags & 4) {  // fractional dosages: the reference's truncating running sum, in sample order
      const SRC* col = src + (long long)blockIdx.x * src_ld;
      aci = 0;
      for (long long i = 0; i < N; ++i) {
        const double g = load_genotype(col, i);
        if (g >= 0.0) aci = (int)((double)aci + g);
      }
    } else {
      aci = (int)ac;
    }
    fill = (an == 0) ? 0.0 : 2.0 * (1.0 * aci / (double)an);
  }
  fill_out[blockIdx.x] = fill;
}

template <typename SRC>
__global__ __launch_bounds__(256) void consolidate_write_kernel(const SRC* __restrict__ src, long long src_ld,
                                                                long long N, long long ld,
                                                                const double* __restrict__ fill_in,
                                                                double* __restrict__ dst) {
  const SRC* col = src + (long long)blockIdx.y * src_ld;
  double* out = dst + (long long)blockIdx.y * ld;
  const double fill = fill_in[blockIdx.y];
  const long long i0 = (long long)blockIdx.x * kConsolChunk;
  const long long i1 = (i0 + kConsolChunk < N) ? i0 + kConsolChunk : N;
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
    const double g = load_genotype(col, i);
    out[i] = (g < 0.0) ? fill : g;
  }
}

// ---- unrelated null models on the device (LinearRegression.cpp:20-69, LogisticRegression.cpp:279-336) ----------------
// One IRLS round: p = 1/(1+exp(-X beta)), V = p(1-p) stored; per-workgroup partial record
//   D = X'VX (d x d), r = X'(y - p) (d), dev = sum y log p + (1-y) log(1-p)      -> lmm_rec_len(d) doubles (last slot unused... dev in slot d*d+d)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(256) void logistic_round_kernel(const double* __restrict__ X, const double* __restrict__ y,
                                                             const double* __restrict__ beta, long long N,
                                                             long long ldx, int d, double* __restrict__ p_out,
                                                             double* __restrict__ v_out, double* __restrict__ partial) {
  extern __shared__ double sm[];
  const int rec = lmm_rec_len(d);
  double* out = partial + (long long)blockIdx.x * rec;
  // pass 1: p and V of this workgroup's samples
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < N; i += 256LL * gridDim.x) {
    double e = 0.0;
    for (int k = 0; k < d; ++k) e += X[i + k * ldx] * beta[k];
    const double p = 1.0 / (1.0 + exp(-e));
    p_out[i] = p;
    v_out[i] = p * (1.0 - p);
  }
  __syncthreads();
  for (int q = 0; q < rec; ++q) {
    int a = 0, b = 0, kind;
    if (q < d * d) {
      kind = 0;
      a = q / d;
      b = q % d;
    } else if (q < d * d + d) {
      kind = 1;
      a = q - d * d;
    } else
      kind = (q == d * d + d) ? 2 : 3;
    double s = 0.0;
    if (kind != 3)
      for (long long i = blockIdx.x * 256LL + threadIdx.x; i < N; i += 256LL * gridDim.x) {
        const double p = p_out[i];
        if (kind == 0)
          s += X[i + a * ldx] * v_out[i] * X[i + b * ldx];
        else if (kind == 1)
          s += X[i + a * ldx] * (y[i] - p);
        else
          s += y[i] * log(p) + (1.0 - y[i]) * log(1.0 - p);
      }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[q] = sm[0];
    __syncthreads();
  }
}
#endif  // RVT_K_ENGINE

// res = y - X beta (linear) and per-workgroup partial sum of res^2
#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(256) void linear_residual_kernel(const double* __restrict__ X, const double* __restrict__ y,
                                                              const double* __restrict__ beta, long long N,
                                                              long long ldx, int d, double* __restrict__ res,
                                                              double* __restrict__ partial) {
  __shared__ double sm[256];
  double s = 0.0;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < N; i += 256LL * gridDim.x) {
    double e = 0.0;
    for (int k = 0; k < d; ++k) e += X[i + k * ldx] * beta[k];
    const double r = y[i] - e;
    res[i] = r;
    s += r * r;
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}
#endif  // RVT_K_ENGINE

// The "null set" the sufficient-statistics kernels read in FamSKAT mode (see the header comment).
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void fam_build_null_kernel(const double* __restrict__ uxy, const double* __restrict__ S,
                                      const double* __restrict__ u1, long long N, long long ld, int d, double sigma2,
                                      double delta, const double* __restrict__ beta, double* __restrict__ Xin,
                                      double* __restrict__ rr, double* __restrict__ v) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double V = sigma2 * (S[i] + delta);  // raw S, as FamSkat.cpp:48-54 uses kinshipS
  double p = 0.0;
  for (int k = 0; k < d; ++k) p += uxy[i + k * N] * beta[k];
  const double r = uxy[i + (long long)d * N] - p;
  for (int k = 0; k < d; ++k) Xin[i + k * ld] = uxy[i + k * N] / V;
  Xin[i + (long long)d * ld] = (u1[i] / fabs(S[i])) / V;
  rr[i] = r / (V * V);
  v[i] = V;
}
#endif  // RVT_K_FAM

// null set of the family MetaCov (MetaCovFamQtl): weights D = 1/((|S| + delta) sigma2), columns [U'X | u1]
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void famcov_build_null_kernel(const double* __restrict__ uxy, const double* __restrict__ S,
                                         const double* __restrict__ u1, long long N, long long ld, int d,
                                         double sigma2, double delta, const double* __restrict__ beta,
                                         double* __restrict__ Xin, double* __restrict__ rr, double* __restrict__ v) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= N) return;
  double p = 0.0;
  for (int k = 0; k < d; ++k) {
    Xin[i + k * ld] = uxy[i + k * N];
    p += uxy[i + k * N] * beta[k];
  }
  const double al = fabs(S[i]);
  const double D = 1.0 / ((al + delta) * sigma2);
  Xin[i + (long long)d * ld] = u1[i];
  Xin[i + (long long)(d + 1) * ld] = (u1[i] / al) / D;  // FastLMM::GetAF numerator: sum u1 ug / |lambda|
  rr[i] = uxy[i + (long long)d * N] - p;                 // uResid = U'y - U'X beta (score statistic)
  v[i] = D;
}
#endif  // RVT_K_FAM

// cmcCollapse / zegginiCollapse (src/Model.cpp:73-89,115-130) of flipped, filtered blocks: gene k owns columns
// [off[k], off[k] + m[k]) of Gp and writes its two collapsed columns to out + (2k) * ld and out + (2k+1) * ld
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void fam_collapse_kernel(const double* __restrict__ Gp, const int* __restrict__ off,
                                    const int* __restrict__ m, long long N, long long ld, double* __restrict__ out) {
  const int k = blockIdx.y;
  const double* g0 = Gp + (long long)off[k] * ld;
  const int mk = m[k];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
    int n = 0;
    for (int j = 0; j < mk; ++j) n += ((int)g0[i + (long long)j * ld] > 0) ? 1 : 0;
    out[i + (long long)(2 * k) * ld] = n > 0 ? 1.0 : 0.0;
    out[i + (long long)(2 * k + 1) * ld] = (double)n;
  }
}
#endif  // RVT_K_FAM

// FastLMM::TestCovariate, SCORE branch (FastLMM.cpp:236-247): stat = U^2 / V, p = chisq_Q(stat, 1) when V > 0
#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ void fam_burden_finish_kernel(const double* __restrict__ cov, int V, const double* __restrict__ ustat,
                                         double* __restrict__ vstat, double* __restrict__ pval) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= V) return;
  const double v = cov[h + (long long)h * V], u = ustat[h];
  vstat[h] = v;
  pval[h] = (v > 0.0) ? chisq_Q(u * u / v, 1.0) : 1.0;
}
#endif  // RVT_K_ENGINE

// raw column sum + monomorphic flag of the columns of one block (MetaCov family mode)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ __launch_bounds__(256) void raw_colstat_kernel(const double* __restrict__ G, long long N, long long ld,
                                                          double* __restrict__ colsum, int* __restrict__ poly) {
  __shared__ double rs[256], rmn[256], rmx[256];
  const double* col = G + (long long)blockIdx.x * ld;
  double s = 0.0, mn = INFINITY, mx = -INFINITY;
  for (long long i = threadIdx.x; i < N; i += 256) {
    const double g = col[i];
    s += g;
    mn = fmin(mn, g);
    mx = fmax(mx, g);
  }
  rs[threadIdx.x] = s;
  rmn[threadIdx.x] = mn;
  rmx[threadIdx.x] = mx;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      rs[threadIdx.x] += rs[threadIdx.x + off];
      rmn[threadIdx.x] = fmin(rmn[threadIdx.x], rmn[threadIdx.x + off]);
      rmx[threadIdx.x] = fmax(rmx[threadIdx.x], rmx[threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    colsum[blockIdx.x] = rs[0];
    poly[blockIdx.x] = (rmn[0] != rmx[0]) ? 1 : 0;
  }
}
#endif  // RVT_K_FAM

// flip / polymorphic decision per genotype column (DataConsolidator.cpp:46-69,94-116): bit 0 = flip, bit 1 = keep
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ __launch_bounds__(256) void fam_colstat_kernel(const double* const* __restrict__ cols, long long N,
                                                          int* __restrict__ flags) {
  __shared__ double rs[256], rmn[256], rmx[256];
  const double* col = cols[blockIdx.x];
  double s = 0.0, mn = INFINITY, mx = -INFINITY;
  bool hard = true;
  for (long long i = threadIdx.x; i < N; i += 256) {
    const double g = col[i];
    s += g;
    mn = fmin(mn, g);
    mx = fmax(mx, g);
    hard = hard && (g == 0.0 || g == 1.0 || g == 2.0);
  }
  const int all_hard = __syncthreads_and(hard ? 1 : 0);
  rs[threadIdx.x] = s;
  rmn[threadIdx.x] = mn;
  rmx[threadIdx.x] = mx;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      rs[threadIdx.x] += rs[threadIdx.x + off];
      rmn[threadIdx.x] = fmin(rmn[threadIdx.x], rmn[threadIdx.x + off]);
      rmx[threadIdx.x] = fmax(rmx[threadIdx.x], rmx[threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)  // bit 0: flip, bit 1: polymorphic, bit 2: every entry is a hard call (0 / 1 / 2)
    flags[blockIdx.x] = (!(rs[0] <= (double)N) ? 1 : 0) | ((rmn[0] != rmx[0]) ? 2 : 0) | (all_hard ? 4 : 0);
}
#endif  // RVT_K_FAM

// kept hard-call columns straight to the int8 plane of the rotation GEMM ([column][ldk] bytes), flipped to 2 - g where
// flagged: what fam_flip_compact_kernel + the column quantiser produce, in one pass over the genotypes
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void fam_flip_quant_kernel(const double* const* __restrict__ src_cols, const int* __restrict__ src_flip,
                                      long long N, long long ldk, signed char* __restrict__ dst) {
  const double* s = src_cols[blockIdx.y];
  const bool fl = src_flip[blockIdx.y] != 0;
  signed char* d = dst + (long long)blockIdx.y * ldk;
  // eight samples per thread: 64 bytes of doubles in, one 8-byte store out (columns start on 128-byte lines, ldk too)
  const long long n8 = (N + 7) / 8;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n8; t += (long long)gridDim.x * blockDim.x) {
    const long long i0 = t * 8;
    unsigned long long w = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int g = (i0 + k < N) ? (int)s[i0 + k] : 0;
      w |= (unsigned long long)(unsigned char)(fl && i0 + k < N ? 2 - g : g) << (8 * k);
    }
    *reinterpret_cast<unsigned long long*>(d + i0) = w;
  }
}
#endif  // RVT_K_FAM

// dst column c (of the compact N x T matrix, leading dimension ld) = kept source column, flipped to 2 - g if flagged
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ void fam_flip_compact_kernel(const double* const* __restrict__ src_cols, const int* __restrict__ src_flip,
                                        long long N, long long ld, double* __restrict__ dst) {
  const double* s = src_cols[blockIdx.y];
  const bool fl = src_flip[blockIdx.y] != 0;
  double* d = dst + (long long)blockIdx.y * ld;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x)
    d[i] = fl ? 2.0 - s[i] : s[i];
}
#endif  // RVT_K_FAM

// Stage A of FamSkat::TestCovariate on the rotated statistics (one workgroup per gene).
//   R = G~' V [G~ | V^-1 U'X | V^-1 u1/|S| | V^-2 r~]  ->  weights, Q, Wm = S - T Cinv T'  (m = M: the block was
//   flipped and filtered before the rotation)
RVT_HD void fam_assemble(const Coop& co, const NullConsts& nc, int M, int Mp, int Cp, const double* parts, int P,
                         GeneScratch ws, GeneStats* out) {
  const int dx = nc.d, d = dx - 1;
  double* R = ws.R;
  for (int idx = co.tid; idx < Mp * Cp; idx += co.nt) {
    const int i = idx / Cp, j = idx % Cp;
    double s = 0.0;
    if ((j >> 4) >= (i >> 4))
      for (int p = 0; p < P; ++p) s += parts[(size_t)p * Mp * Cp + idx];
    R[idx] = s;
  }
  co.sync();
  double qpart = 0.0;
  for (int i = co.tid; i < M; i += co.nt) {
    // FastGetAF (FastLMM.cpp:402-443): 0.5 * alpha.g / denom; denom == 0 -> 0
    const double af = (nc.rss == 0.0) ? 0.0 : 0.5 * (R[(size_t)i * Cp + M + d] / nc.rss);
    const double w = beta_density(af, 1.0, 25.0);  // FamSkat.cpp:129-137: beta1 = 1, beta2 = 25 always
    const double u = R[(size_t)i * Cp + M + d + 1];
    ws.bw[i] = w;
    ws.bw[Mp + i] = NAN;  // never equal to the SKAT weights: the eigen stage builds SKAT's own matrix
    qpart += (w * u) * (w * u);
  }
  const double Q = co.sum(qpart);
  double* TC = ws.vecs;  // M x d: T Cinv
  for (int idx = co.tid; idx < M * d; idx += co.nt) {
    const int i = idx / d, b = idx % d;
    double s = 0.0;
    for (int a = 0; a < d; ++a) s += R[(size_t)i * Cp + M + a] * nc.Cinv[a * dx + b];
    TC[idx] = s;
  }
  co.sync();
  for (int idx = co.tid; idx < M * M; idx += co.nt) {
    const int i = idx % M, j = idx / M;
    const double S = (j >= i) ? R[(size_t)i * Cp + j] : R[(size_t)j * Cp + i];
    double q = 0.0;
    for (int b = 0; b < d; ++b) q += TC[(size_t)i * d + b] * R[(size_t)j * Cp + M + b];
    ws.Wm[idx] = S - q;
  }
  if (co.tid == 0) {
    out->status = 0;
    out->n_variants = M;
    out->n_poly = M;
    out->flip_count = 0;
    out->skat_Q = Q;
    out->skat_nlambda = 0;
    out->skat_lambda_off = 0;
    out->zimz_lambda_off = M;
    out->zimz_nlambda = 0;
    out->skato_ok = 0;
    out->skato_single = 0;
    out->cmc_ok = 0;
    out->zeg_ok = 0;
  }
  co.sync();
}

#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(1024) void fam_assemble_kernel(const GeneDesc* __restrict__ genes,
                                                           const NullConsts* __restrict__ ncp) {
  __shared__ double red[64];
  __shared__ NullConsts nc;
  const GeneDesc gd = genes[blockIdx.x];
  if (threadIdx.x == 0) nc = *ncp;
  __syncthreads();
  Coop co{(int)threadIdx.x, (int)blockDim.x, red};
  GeneScratch ws = gene_scratch_carve(gd.scratch, gd.Mp, gd.Cp);
  fam_assemble(co, nc, gd.M, gd.Mp, gd.Cp, gd.parts, gd.n_wparts, ws, gd.stats);
}
#endif  // RVT_K_ENGINE

}  // namespace rvt
