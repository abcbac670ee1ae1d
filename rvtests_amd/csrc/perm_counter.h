// rvtests_amd — counter-based permutations for the SKAT permutation p-value (SkatTest::fit, src/Model.h:2706-2718;
// Permutation.h:69-98) when the run is NOT tied to the reference's own rand() stream ("sharded mode uses a
// per-gene counter-based RNG, statistical parity").
//
// The exact mode (perm_kernels.hip.h) replays glibc's rand(): one process-wide stream in gene order, Fisher-Yates with
// dependent random swaps — sequential across genes and bound by random DRAM accesses (~3 k shuffles/s at N = 500 000).
// Here shuffle s of gene g is a KEYED BIJECTION pi_{g,s} of [0, N): an unbalanced Feistel network over ceil(log2 N) bits
// with six keyed rounds, cycle-walked into [0, N) (apply again while the image is >= N: a bijection of the power-of-two
// domain restricted this way is a bijection of [0, N)).  The round keys come from Philox4x32-10 with the counter
// (shuffle, gene id) and the key (seed): any engine context, any device, any order of genes gives the same permutations.
// pi(i) is evaluated per element, so
//   * nothing is stored per shuffle (the exact mode keeps an N x B index array and applies it with scattered accesses);
//   * the permuted residual r[pi(i)] is a gather from ONE N-vector (4 MB at N = 500 000: resident in L2), consumed in
//     registers as the A operand of the fp64 matrix instruction that forms R_pi' G (perm_counter_q_kernel) — the
//     permuted residual matrix of a chunk (8 GB for 2048 shuffles) is never written.
// Statistical parity, not replay: the permutation p-value estimates the same tail probability as the reference's, with
// the same adaptive stopping rule; tests compare the two modes within binomial error.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RVT_PC_HD __host__ __device__ inline
#else
#define RVT_PC_HD inline
#endif

namespace rvt {

constexpr int kPermRounds = 6;

struct PermKeys {
  uint32_t k[kPermRounds];
};

RVT_PC_HD void philox4x32_10(const uint32_t (&ctr)[4], const uint32_t (&key)[2], uint32_t (&out)[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1,
                   n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// round keys of shuffle `shuffle` of gene `gene` under `seed`
RVT_PC_HD PermKeys perm_keys(uint64_t seed, uint64_t gene, uint32_t shuffle) {
  PermKeys pk;
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t o[4];
  const uint32_t c0[4] = {shuffle, 0u, (uint32_t)gene, (uint32_t)(gene >> 32)};
  philox4x32_10(c0, key, o);
  pk.k[0] = o[0];
  pk.k[1] = o[1];
  pk.k[2] = o[2];
  pk.k[3] = o[3];
  const uint32_t c1[4] = {shuffle, 1u, (uint32_t)gene, (uint32_t)(gene >> 32)};
  philox4x32_10(c1, key, o);
  pk.k[4] = o[0];
  pk.k[5] = o[1];
  return pk;
}

RVT_PC_HD uint32_t perm_mix(uint32_t x) {  // murmur3's 32-bit finaliser: full avalanche
  x ^= x >> 16;
  x *= 0x85ebca6bu;
  x ^= x >> 13;
  x *= 0xc2b2ae35u;
  x ^= x >> 16;
  return x;
}

RVT_PC_HD int perm_bits(uint64_t n) {  // ceil(log2 n), at least 2
  int b = 2;
  while (((uint64_t)1 << b) < n) ++b;
  return b;
}

// pi(i) for i in [0, n), n <= 2^31
RVT_PC_HD uint32_t perm_index(uint32_t i, uint32_t n, int bits, const PermKeys& pk) {
  const int hb = bits >> 1, lb = bits - hb;
  const uint32_t lmask = (1u << lb) - 1u, hmask = (1u << hb) - 1u;
  uint32_t x = i;
  do {
    uint32_t L = x >> lb, R = x & lmask;
    for (int r = 0; r < kPermRounds; r += 2) {
      L = (L ^ perm_mix(R + pk.k[r])) & hmask;
      R = (R ^ perm_mix(L + pk.k[r + 1])) & lmask;
    }
    x = (L << lb) | R;
  } while (x >= n);
  return x;
}

}  // namespace rvt
