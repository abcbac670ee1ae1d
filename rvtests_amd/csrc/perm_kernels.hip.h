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

namespace rvt {

// idx[k * B + p] = k : permutation-minor layout, so that the sequential side of the swaps is coalesced
__global__ void perm_init_kernel(uint32_t* __restrict__ idx, long long N, int B) {
  const long long n = N * (long long)B;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    idx[i] = (uint32_t)(i / B);
}

// One thread = one permutation.  states: B x 31 words, the generator state at the start of each shuffle, ordered
// oldest word first (x[t] = o[k-31+t]); a draw is x[t] += x[(t+28) % 31] with t cycling 0..30.
__global__ __launch_bounds__(64) void perm_fisher_yates_kernel(const uint32_t* __restrict__ states,
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

// std::random_shuffle as libstdc++ implements it (the KBAC permutations, regression/kbac.cpp:323): i = 1 .. N-1,
// j = rand() % (i + 1), swap(i, j).  Same state layout and draw as perm_fisher_yates_kernel, forward order.
__global__ __launch_bounds__(64) void perm_random_shuffle_kernel(const uint32_t* __restrict__ states,
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

// next[k] = cur[idx[k][p]] for a vector of bytes (the 0 / 1 phenotype of the KBAC permutations)
__global__ void perm_apply_u8_kernel(const uint32_t* __restrict__ idx, const unsigned char* __restrict__ cur,
                                     unsigned char* __restrict__ next, long long N, int B, int p) {
  const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (k >= N) return;
  next[k] = cur[idx[k * B + p]];
}

// out[c] = vec[carrier[c]]
__global__ void perm_gather_u8_kernel(const unsigned char* __restrict__ vec, const int* __restrict__ carrier, int n,
                                      unsigned char* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) out[c] = vec[carrier[c]];
}

// KBAC genotype-pattern id of every sample (regression/kbac.cpp:120-147), exactly as the reference's double arithmetic
// runs: columns in order, invalid codings (anything but 0 / 1 / 2: imputed means) count as wild type, p3[k] = the host's
// pow(3.0, k).  G: flipped / polymorphic block (column-major, ld), cols: the n_used columns that survive the frequency trim.
__global__ void kbac_pattern_kernel(const double* __restrict__ G, long long N, long long ld, const int* __restrict__ cols,
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

// cumulative application of shuffle p to the current residual vector: next[k] = cur[idx[k][p]];
// also column p of the chunk matrix Rp (N x B column-major: Rp[k + p*N])
__global__ void perm_apply_kernel(const uint32_t* __restrict__ idx, const double* __restrict__ cur,
                                  double* __restrict__ next, double* __restrict__ Rp, long long N, int B, int p) {
  const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (k >= N) return;
  const double v = cur[idx[k * B + p]];
  next[k] = v;
  Rp[k + (long long)p * N] = v;
}

// Small sample counts: C[p + j*B] = sum_i G[i + j*ld] Rp[i + p*N] summed in sample order i = 0 .. N-1, exactly as the
// reference's (and the oracle's) dot product runs.  With a handful of samples — the reference's own example has 9 —
// many shuffles reproduce the observed Q mathematically, and whether such a tie counts as "greater" is decided by the
// last bit; only the same summation order resolves it the same way.
__global__ void perm_dot_sequential_kernel(const double* __restrict__ Rp, const double* __restrict__ G, long long N,
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

// Q_p = sum_j w_j (g_j . r_p)^2 from C = Rp * G' (B x m, column-major, ldc = B); bw[j] = sqrt(w_j)
__global__ void perm_q_kernel(const double* __restrict__ C, const double* __restrict__ bw, int B, int m,
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

}  // namespace rvt
