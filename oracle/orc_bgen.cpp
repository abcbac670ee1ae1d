// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_api.h).  CPU restatement of the BGEN genotype-probability front end:
//   * layout 1 (v1.1) block: 3 x uint16 per sample / 32768, missing = all three zero       BGenFile::parseLayout1  libBgen/BGenFile.cpp:205-238
//   * layout 2 (v1.2 / 1.3) block: N, K, ploidy range, ploidy/missing bytes, phased flag, B bits, packed probabilities
//                                                                                           BGenFile::parseLayout2  libBgen/BGenFile.cpp:321-392
//   * B-bit little-endian bit stream -> float(v) * scale, scale = float(1.0 / float(2^B - 1)) BitReader              libBgen/BitReader.h:14-72
//   * stored probabilities per sample (phased: per haplotype K-1 values + the remainder; unphased: C(Z+K-1, K-1) - 1
//     values + the remainder; remainder = 1.0f minus the values, float, in order)           libBgen/BGenFile.cpp:354-384
//   * probabilities -> the genotype the association tests see (always a dosage)              BGenGenotypeExtractor::getGenotype  src/BGenGenotypeExtractor.cpp:413-478
// The reference holds the UNCOMPRESSED block after uncompress / ZSTD_decompress (BGenFile.cpp:289-319); this restatement
// starts there.  Pinned against the reference's own golden files libBgen/test/*.bgen + *.vcf.correct (the %g-printed
// probabilities of BGenVariant::printGP / printHP, libBgen/BGenVariant.cpp:172-240) in tests/test_bgen_cpu.py; the
// reference's BGenFile.cpp itself needs zstd and sqlite headers the image lacks and is not compiled.
// Quirks kept: getGenotype reads prob[index + 1] and prob[index + 2] whatever the sample's ploidy and phasing, so a
// phased diploid sample gets (1 - p_hap1) + 2 p_hap2 and a haploid biallelic sample reads the NEXT sample's first
// probability (past the end of the vector for the last sample: 0 here, undefined in the reference).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {
const double kMissingGenotype = -9.0;  // libVcf/VCFConstant.h:4 (MISSING_GENOTYPE)

// BGenFile::choose (libBgen/BGenFile.cpp:438-453), int arithmetic
int choose(int n, int m) {
  if (m == 1) return n;
  if (n == 1) return 1;
  int ret = 1;
  for (int i = 0; i < m; ++i) ret *= (n - i);
  for (int i = 0; i < m; ++i) ret /= (i + 1);
  return ret;
}

struct BitReader {  // libBgen/BitReader.h
  const uint8_t* data;
  int64_t offset, len;
  unsigned availableBits;
  int B;
  uint64_t value, mask;
  float scale;
  BitReader(const uint8_t* d, int64_t l, int b) : data(d), offset(0), len(l), availableBits(0), B(b), value(0) {
    mask = (1ull << B) - 1;  // (the reference's `(1 << B) - 1` on an int: the same bits for B <= 31 with wrap-around;
                             //  its own golden file complex.31bits.bgen relies on that)
    scale = 1.0f;
    for (int i = 0; i < B; ++i) scale *= 2;
    scale -= 1;
    scale = (float)(1.0 / scale);
  }
  float next() {
    if (B == 8) return (float)data[offset++] * scale;
    if (B == 16) {
      uint16_t v;
      std::memcpy(&v, data + offset, 2);
      offset += 2;
      return (float)v * scale;
    }
    if (B == 32) {
      uint32_t v;
      std::memcpy(&v, data + offset, 4);
      offset += 4;
      return (float)v * scale;
    }
    while (availableBits < (unsigned)B && offset < len) {
      value |= ((uint64_t)data[offset]) << availableBits;
      offset++;
      availableBits += 8;
    }
    const float res = (float)(value & mask);
    availableBits -= B;
    value >>= B;
    return res * scale;
  }
};
}  // namespace

extern "C" {

// Decode one uncompressed probability block.  missing / ploidy: N bytes; index: N + 1; prob: capacity prob_cap floats.
// info[0] = phased, info[1] = B, info[2] = K.  Returns the number of stored probabilities, or
//   -1 malformed / sample count differs, -2 prob_cap too small
int64_t orc_bgen_decode(const uint8_t* blk, int64_t len, int layout, int64_t N, int* info, uint8_t* missing,
                        uint8_t* ploidy, int64_t* index, float* prob, int64_t prob_cap) {
  if (layout == 1) {
    if (len < 6 * N) return -1;
    if (prob_cap < 3 * N) return -2;
    for (int64_t i = 0; i < N; ++i) {
      uint16_t v[3];
      std::memcpy(v, blk + 6 * i, 6);
      float p[3];
      for (int k = 0; k < 3; ++k) p[k] = (float)v[k] / 32768;
      ploidy[i] = 2;
      index[i] = 3 * i;
      missing[i] = (p[0] == 0 && p[1] == 0 && p[2] == 0) ? 1 : 0;
      for (int k = 0; k < 3; ++k) prob[3 * i + k] = p[k];
    }
    index[N] = 3 * N;
    if (info) {
      info[0] = 0;
      info[1] = 16;
      info[2] = 2;
    }
    return 3 * N;
  }
  if (layout != 2 || len < 10 + N) return -1;
  uint32_t nIndv;
  uint16_t K;
  std::memcpy(&nIndv, blk, 4);
  std::memcpy(&K, blk + 4, 2);
  if ((int64_t)nIndv != N) return -1;
  const uint8_t* pm = blk + 8;
  const int phased = blk[8 + N] != 0;
  const int B = blk[8 + N + 1];
  if (B < 1 || B > 32) return -1;
  if (info) {
    info[0] = phased;
    info[1] = B;
    info[2] = K;
  }
  BitReader br(blk + 8 + N + 2, len - 8 - N - 2, B);
  int64_t np = 0;
  for (int64_t i = 0; i < N; ++i) {
    index[i] = np;
    const int Z = pm[i] & 0x3f;
    ploidy[i] = (uint8_t)Z;
    missing[i] = (pm[i] & 0x80) ? 1 : 0;
    if (phased) {
      for (int j = 0; j < Z; ++j) {
        float remain = 1.0f;
        for (int k = 0; k < K - 1; ++k) {
          const float p = br.next();
          if (np >= prob_cap) return -2;
          prob[np++] = p;
          remain -= p;
        }
        if (np >= prob_cap) return -2;
        prob[np++] = remain;
      }
    } else {
      const int nc = choose(Z + K - 1, K - 1);
      float remain = 1.0f;
      for (int j = 0; j < nc - 1; ++j) {
        const float p = br.next();
        if (np >= prob_cap) return -2;
        prob[np++] = p;
        remain -= p;
      }
      if (np >= prob_cap) return -2;
      prob[np++] = remain;
    }
  }
  index[N] = np;
  return np;
}

// BGenGenotypeExtractor::getGenotype for sample i of a decoded block (K = number of alleles)
double orc_bgen_genotype(int K, const uint8_t* missing, const uint8_t* ploidy, const int64_t* index, const float* prob,
                         int64_t n_prob, int64_t i) {
  if (missing[i]) return kMissingGenotype;
  if (ploidy[i] != 2 && ploidy[i] != 1) return kMissingGenotype;
  auto P = [&](int64_t q) -> float { return q < n_prob ? prob[q] : 0.0f; };
  const int64_t b = index[i];
  if (K == 2) return P(b + 1) + P(b + 2) * 2.0;
  if (K == 1) return 2;
  const double total = P(b) + P(b + 1) + P(b + 2);  // float additions, as the reference's expression
  if (total > 0.) return (P(b + 1) + P(b + 2) * 2.0) / total;
  return kMissingGenotype;
}

// block -> genotype per file sample (out[N]); returns 0 or orc_bgen_decode's error
int orc_bgen_block_genotypes(const uint8_t* blk, int64_t len, int layout, int64_t N, double* out) {
  std::vector<uint8_t> missing(N), ploidy(N);
  std::vector<int64_t> index(N + 1);
  int64_t cap = 3 * N + 16;
  std::vector<float> prob;
  int info[3];
  int64_t np;
  for (;;) {
    prob.resize(cap);
    np = orc_bgen_decode(blk, len, layout, N, info, missing.data(), ploidy.data(), index.data(), prob.data(), cap);
    if (np != -2) break;
    cap *= 4;
    if (cap > (1ll << 31)) return -2;
  }
  if (np < 0) return (int)np;
  for (int64_t i = 0; i < N; ++i)
    out[i] = orc_bgen_genotype(info[2], missing.data(), ploidy.data(), index.data(), prob.data(), np, i);
  return 0;
}

}  // extern "C"
