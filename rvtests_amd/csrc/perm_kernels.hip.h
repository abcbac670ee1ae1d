// rvtests_amd — SKAT permutation p-values (SkatTest::fit, src/Model.h:2706-2718) with the reference's own random
// stream.
//
// The reference shuffles the residual vector cumulatively with Fisher–Yates driven by glibc rand()
// (src/LinearAlgebra.h:8-21: for i = N-1 .. 1: j = rand() % (i+1); swap(v[i], v[j])) and recomputes
// Q = || K_sqrt r_perm ||^2 (Skat.cpp:107-116) until numX + numEqual >= 2 nPerm alpha or nPerm permutations are done
// (src/Permutation.h:69-98).  glibc's default generator (TYPE_3) is the additive lagged-Fibonacci recurrence
// o[k] = o[k-31] + o[k-3] (mod 2^32), output o[k] >> 1 — LINEAR, so the state after the N-1 draws of one shuffle is
// J s with J = A^(N-1) a 31 x 31 matrix over Z/2^32.  The host jumps from permutation to permutation with J; on the
// device every permutation of a chunk is generated INDEPENDENTLY by one thread (its own generator state, its own
// index array), then the shuffles are composed in order (they are cumulative), the permuted residuals of the chunk
// form an N x B matrix and all Q of the chunk come from ONE integer-plane product (rot_gemm.hip.h) with the flipped /
// filtered genotype block.
// The permutations are therefore exactly the reference's; Q is evaluated in fp64 where the reference uses fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "perm_counter.h"

namespace rvt {

// idx[k * B + p] = k : permutation-minor layout, so that the sequential side of the swaps is coalesced
#if !defined(RVT_K_SPLIT) || defined(RVT_K_PERM)
static __global__ void perm_init_kernel(uint32_t* __restrict__ idx, long long N, int B) {
  const long long n = N * (long long)B;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    idx[i] = (uint32_t)(i / B);
}
#endif  // RVT_K_PERM

// One thread = one permutation.  states: B x 31 words, the generator state at the start of each shuffle, ordered
// oldest word first (x[t] = o[k-31+t]); a draw is x[t] += x[(t+28) % 31] with t cycling 0..30.
#if !defined(RVT_K_SPLIT) || defined(RVT_K_PERM)
static __global__ __launch_bounds__(64) void perm_fisher_yates_kernel(const uint32_t* __restrict__ states,
                                                               uint32_t* __restrict__ idx, long long N, int B) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  uint32_t x[31];
#pragma unroll
  for (int t = 0; t < 31; ++t) x[t] = states[(long long)p * 31 + t];
  long long i = N - 1;
  while (i >= 1) {
#pragma unroll
    for (int t = 0; t < 31; ++t) {
      if (i >= 1) {
        const uint32_t v = x[t] + x[(t + 28) % 31];
        x[t] = v;
        const uint32_t r = v >> 1;                          // rand()
        const uint32_t j = r % (uint32_t)(i + 1);           // 0 <= j <= i
        if ((long long)j != i) {
          uint32_t* pi = idx + i * (long long)B + p;
          uint32_t* pj = idx + (long long)j * B + p;
          const uint32_t a = *pi, b = *pj;
          *pi = b;
          *pj = a;
        }
        --i;
      }
    }
  }
}
#endif  // RVT_K_PERM

// std::random_shuffle as libstdc++ implements it (the KBAC permutations, regression/kbac.cpp:323): i = 1 .. N-1,
// j = rand() % (i + 1), swap(i, j).  Same state layout and draw as perm_fisher_yates_kernel, forward order.
#if !defined(RVT_K_SPLIT) || defined(RVT_K_PERM)
static __global__ __launch_bounds__(64) void perm_random_shuffle_kernel(const uint32_t* __restrict__ states,
                                                                 uint32_t* __restrict__ idx, long long N, int B) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  uint32_t x[31];
#pragma unroll
  for (int t = 0; t < 31; ++t) x[t] = states[(long long)p * 31 + t];
  long long i = 1;
  while (i < N) {
#pragma unroll
    for (int t = 0; t < 31; ++t) {
      if (i < N) {
        const uint32_t v = x[t] + x[(t + 28) % 31];
        x[t] = v;
        const uint32_t r = v >> 1;                          // rand()
        const uint32_t j = r % (uint32_t)(i + 1);           // 0 <= j <= i
        if ((long long)j != i) {
          uint32_t* pi = idx + i * (long long)B + p;
          uint32_t* pj = idx + (long long)j * B + p;
          const uint32_t a = *pi, b = *pj;
          *pi = b;
          *pj = a;
        }
        ++i;
      }
    }
  }
}
#endif  // RVT_K_PERM

// next[k] = cur[idx[k][p]] for a vector of bytes (the 0 / 1 phenotype of the KBAC permutations)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_PERM)
static __global__ void perm_apply_u8_kernel(const uint32_t* __restrict__ idx, const unsigned char* __restrict__ cur,
                                     unsigned char* __restrict__ next, long long N, int B, int p) {
  const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (k >= N) return;
  next[k] = cur[idx[k * B + p]];
}
#endif  // RVT_K_PERM

// out[c] = vec[carrier[c]]
#if !defined(RVT_K_SPLIT) || defined(RVT_K_PERM)
static __global__ void perm_gather_u8_kernel(const unsigned char* __restrict__ vec, const int* __restrict__ carrier, int n,
                                      unsigned char* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) out[c] = vec[carrier[c]];
}
#endif  // RVT_K_PERM

// KBAC genotype-pattern id of every sample (regression/kbac.cpp:120-147), exactly as the reference's double arithmetic
// runs: columns in order, invalid codings (anything but 0 / 1 / 2: imputed means) count as wild type, p3[k] = the host's
// pow(3.0, k).  G: flipped / polymorphic block (column-major, ld), cols: the n_used columns that survive the frequency trim.
#if !defined(RVT_K_SPLIT) || defined(RVT_K_PERM)
static __global__ void kbac_pattern_kernel(const double* __restrict__ G, long long N, long long ld, const int* __restrict__ cols,
                                    int n_used, const double* __restrict__ p3, double* __restrict__ id) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double ixiix = 3486784401.0;  // pow(9.0, 10.0)
  double L = 0.0, R = 0.0;
  unsigned lastCnt = 0, tmpCnt = 0;
  for (unsigned j = 0; j != (unsigned)n_used; ++j) {
    double g = G[(long long)cols[j] * ld + i];
    if (g != 0.0 && g != 1.0 && g != 2.0) g = 0.0;
    if (g == 0.0) continue;
    R = __dadd_rn(R, __dmul_rn(p3[j - lastCnt], g));  // (no contraction: two roundings, as on the host)
    if (R >= ixiix) {
      L = L + 1.0;
      R = R - ixiix;
      lastCnt = lastCnt + tmpCnt + 1;
      tmpCnt = 0;
    } else {
      ++tmpCnt;
    }
  }
  id[i] = __dadd_rn(L, __dmul_rn(R, 1e-10));
}
#endif  // RVT_K_PERM

// cumulative application of shuffle p to the current residual vector: next[k] = cur[idx[k][p]];
// also column p of the chunk matrix Rp (N x B column-major: Rp[k + p*N])
#if !defined(RVT_K_SPLIT) || defined(RVT_K_PERM)
static __global__ void perm_apply_kernel(const uint32_t* __restrict__ idx, const double* __restrict__ cur,
                                  double* __restrict__ next, double* __restrict__ Rp, long long N, int B, int p) {
  const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (k >= N) return;
  const double v = cur[idx[k * B + p]];
  next[k] = v;
  Rp[k + (long long)p * N] = v;
}
#endif  // RVT_K_PERM

// Small sample counts: C[p + j*B] = sum_i G[i + j*ld] Rp[i + p*N] summed in sample order i = 0 .. N-1, exactly as the
// reference's (and the oracle's) dot product runs.  With a handful of samples — the reference's own example has 9 —
// many shuffles reproduce the observed Q mathematically, and whether such a tie counts as "greater" is decided by the
// last bit; only the same summation order resolves it the same way.
#if !defined(RVT_K_SPLIT) || defined(RVT_K_PERM)
static __global__ void perm_dot_sequential_kernel(const double* __restrict__ Rp, const double* __restrict__ G, long long N,
                                           long long ld, int nb, int m, int B, double* __restrict__ C) {
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t >= (long long)nb * m) return;
  const int p = (int)(t % nb), j = (int)(t / nb);
  const double* r = Rp + (long long)p * N;
  const double* g = G + (long long)j * ld;
  double s = 0.0;
  for (long long i = 0; i < N; ++i) s += g[i] * r[i];
  C[p + (long long)j * B] = s;
}
#endif  // RVT_K_PERM

// Q_p = sum_j w_j (g_j . r_p)^2 from C = Rp * G' (B x m, column-major, ldc = B); bw[j] = sqrt(w_j)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_PERM)
static __global__ void perm_q_kernel(const double* __restrict__ C, const double* __restrict__ bw, int B, int m,
                              double* __restrict__ Q) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B) return;
  double s = 0.0;
  for (int j = 0; j < m; ++j) {
    const double u = bw[j] * C[p + (long long)j * B];
    s += u * u;
  }
  Q[p] = s;
}
#endif  // RVT_K_PERM

// =====================================================================================================================
// Counter-based mode (perm_counter.h): C = R_pi' G for a chunk of shuffles without ever storing a permutation or a
// permuted residual vector.
//   grid (sample slices, ceil(n_shuffles / 64)), one wave per workgroup: the wave owns 64 shuffles (4 row tiles of the
//   fp64 matrix instruction) and, per pass, up to 64 variants (4 column tiles); it walks its sample slice in steps of 16
//   samples.  A operand (16 shuffles x 4 samples) = r[pi_s(i)], one permutation evaluation and one gather from the
//   L2-resident residual vector per lane; B operand (4 samples x 16 variants) = the flipped, polymorphic genotype block,
//   each lane reading 4 consecutive samples (32 B) of its column per group of four steps, as the sufficient-statistics
//   kernels do.  Partial products go to part[slice][shuffle][variant]; perm_counter_q_kernel adds the slices in a fixed
//   order and forms Q = sum_j (w_j^1/2 C_sj)^2 (Skat.cpp:107-116).
// =====================================================================================================================
typedef double pc_d4_t __attribute__((ext_vector_type(4)));

#if !defined(RVT_K_SPLIT) || defined(RVT_K_PERM)
static __global__ __launch_bounds__(64, 2) void perm_counter_partial_kernel(
    const double* __restrict__ G, long long ld, long long N, int m, const double* __restrict__ res,
    unsigned long long seed, unsigned long long gene, unsigned shuffle0, int n_shuffles, int groups_per_slice,
    int Mp, double* __restrict__ part) {
  const int lane = threadIdx.x & 63, v = lane & 15, k = lane >> 4;
  const int slice = blockIdx.x, bt = blockIdx.y;
  const long long ngroups = (N + 15) >> 4;  // groups of 16 samples
  const long long g0 = (long long)slice * groups_per_slice;
  long long g1 = g0 + groups_per_slice;
  if (g1 > ngroups) g1 = ngroups;
  const int bits = perm_bits((unsigned long long)N);
  PermKeys pk[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) pk[rt] = perm_keys(seed, gene, shuffle0 + (unsigned)(bt * 64 + rt * 16 + v));
  for (int c0 = 0; c0 < Mp; c0 += 64) {  // (genes wider than 64 variants: the permutations are evaluated once per pass)
    pc_d4_t acc[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = pc_d4_t{0.0, 0.0, 0.0, 0.0};
    const double* col[4];
    bool colok[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int j = c0 + ct * 16 + v;
      colok[ct] = j < m;
      col[ct] = G + (long long)(colok[ct] ? j : 0) * ld;
    }
    for (long long g = g0; g < g1; ++g) {
      const long long i0 = g * 16 + (long long)k * 4;  // this lane's four consecutive samples
      double b[4][4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int t = 0; t < 4; ++t) b[ct][t] = (colok[ct] && i0 + t < N) ? col[ct][i0 + t] : 0.0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const long long i = i0 + t;
        const bool in = i < N;
        double a[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const uint32_t src = perm_index((uint32_t)(in ? i : 0), (uint32_t)N, bits, pk[rt]);
          a[rt] = in ? res[src] : 0.0;
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
            acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt], b[ct][t], acc[rt][ct], 0, 0, 0);
      }
    }
    // D: row (shuffle) = rt * 16 + k + 4 i, column (variant) = ct * 16 + v
    double* out = part + ((long long)slice * n_shuffles) * Mp;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int srow = bt * 64 + rt * 16 + k + 4 * i;
        if (srow < n_shuffles) {
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            const int j = c0 + ct * 16 + v;
            if (j < Mp) out[(long long)srow * Mp + j] = acc[rt][ct][i];
          }
        }
      }
  }
}
#endif  // RVT_K_PERM

// Q[s] = sum_j bw_j^2 (sum_slices part[slice][s][j])^2, slices in order
#if !defined(RVT_K_SPLIT) || defined(RVT_K_PERM)
static __global__ void perm_counter_q_kernel(const double* __restrict__ part, int n_slices, int n_shuffles, int Mp, int m,
                                      const double* __restrict__ bw, double* __restrict__ Q) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_shuffles) return;
  double q = 0.0;
  for (int j = 0; j < m; ++j) {
    double c = 0.0;
    for (int sl = 0; sl < n_slices; ++sl) c += part[((long long)sl * n_shuffles + s) * Mp + j];
    const double w = bw[j] * c;
    q += w * w;
  }
  Q[s] = q;
}
#endif  // RVT_K_PERM

}  // namespace rvt
