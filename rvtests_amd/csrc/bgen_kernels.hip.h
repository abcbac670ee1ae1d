// rvtests_amd — BGEN genotype-probability blocks -> the genotype (dosage) column the association tests read.
//
// Replaces, for the UNCOMPRESSED probability block of a variant (what the reference holds after uncompress /
// ZSTD_decompress, libBgen/BGenFile.cpp:289-319):
//   BGenFile::parseLayout1                 libBgen/BGenFile.cpp:205-238   v1.1: 3 x uint16 / 32768, all-zero = missing
//   BGenFile::parseLayout2 + BitReader     libBgen/BGenFile.cpp:321-392, libBgen/BitReader.h:14-72
//                                          v1.2 / 1.3: ploidy / missing bytes, phased flag, B-bit little-endian values,
//                                          float(v) * scale, the remainder 1.0f - sum in float, in order
//   BGenGenotypeExtractor::getGenotype     src/BGenGenotypeExtractor.cpp:413-478   prob[index + 1] + 2 prob[index + 2] ...
// with the reference's arithmetic type at every step (float products and differences, the final sum in double), so the
// doubles written are the reference's bit for bit (tests/test_gpu_bgen.py: against the CPU restatement, itself pinned to
// the reference's golden files libBgen/test/*.bgen + *.vcf.correct).
//
// A sample's values start at a bit offset that depends on the ploidies of all samples before it (and the number of
// alleles), so a block is decoded in three passes like the VCF text (vcf_kernels.hip.h):
//   1. bgen_count_kernel   values per 256-sample segment                              grid (segments, variants)
//   2. bgen_scan_kernel    exclusive scan over the segments of a variant              grid (variants)
//   3. bgen_decode_kernel  scan inside the segment, fetch, convert, write the double  grid (segments, variants)
// Layout 1 needs no scan (6 bytes per sample) and goes through pass 3 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rvt {

constexpr int kBgenSeg = 256;           // samples per segment (one per thread)
constexpr double kBgenMissing = -9.0;   // MISSING_GENOTYPE (libVcf/VCFConstant.h:4)

struct BgenRecord {
  long long off;  // byte offset of the block in the staging buffer (4-byte aligned, >= 16 readable bytes behind its end)
  long long len;
  int layout;     // 1 or 2
  int K;          // alleles (layout 1: 2)
  int phased;
  int bits;       // B
  float scale;    // BitReader's scale: float(1.0 / float(2^B - 1)), computed by the host in float
  int zmax;       // largest ploidy the block header declares (byte 7): a sample above it raises the error flag
  int alt;        // multi-allelic mode: the alternative allele asked for; > 1 -> every sample missing
                  // (BGenGenotypeExtractor::getGenotypeForAltAllele, src/BGenGenotypeExtractor.cpp:470-482)
};

// BGenFile::choose (libBgen/BGenFile.cpp:438-453), int arithmetic
__device__ __forceinline__ int bgen_choose(int n, int m) {
  if (m == 1) return n;
  if (n == 1) return 1;
  int ret = 1;
  for (int i = 0; i < m; ++i) ret *= (n - i);
  for (int i = 0; i < m; ++i) ret /= (i + 1);
  return ret;
}
// probabilities the reference STORES for a sample of ploidy Z (including the remainders) and the values it READS
__device__ __forceinline__ int bgen_stored(int Z, int K, int phased) { return phased ? Z * K : bgen_choose(Z + K - 1, K - 1); }
__device__ __forceinline__ int bgen_values(int Z, int K, int phased) {
  return phased ? Z * (K - 1) : bgen_choose(Z + K - 1, K - 1) - 1;
}

// float product / difference / sum rounded on their own, as the reference's separate statements are (the HIP wrappers
// __fmul_rn / __fsub_rn are plain operators the compiler may fuse into one fma)
__device__ __forceinline__ float bgen_mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float bgen_sub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ float bgen_add(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}

// value q of the packed area (B bits each, little-endian bit order) -> float(v) * scale
__device__ __forceinline__ float bgen_value(const unsigned char* __restrict__ packed, long long q, int bits, float scale) {
  const long long bit = q * bits;
  const long long byte = bit >> 3;
  const unsigned* w = reinterpret_cast<const unsigned*>(packed + (byte & ~3ll));  // (packed is 4-byte aligned)
  const unsigned long long win = ((unsigned long long)w[1] << 32) | w[0];
  const int sh = (int)((byte & 3) * 8 + (bit & 7));  // <= 31: 33 bits of the window are left
  const unsigned long long mask = (bits >= 32) ? 0xFFFFFFFFull : ((1ull << bits) - 1);
  const unsigned v = (unsigned)((win >> sh) & mask);
  return bgen_mul(__uint2float_rn(v), scale);
}

// stored probability t (0-based, counted from sample i's first stored probability, running on into the following
// samples exactly as the reference's flat prob vector does); rbase = index of sample i's first packed value
__device__ __forceinline__ float bgen_stored_prob(const unsigned char* __restrict__ pm, const unsigned char* __restrict__ packed,
                                                  long long N, int K, int phased, int bits, float scale, long long i,
                                                  long long rbase, int t) {
  int nst = 0;
  for (;;) {
    if (i >= N) return 0.0f;  // past the end of the reference's vector (undefined there)
    const int Z = pm[i] & 0x3f;
    nst = bgen_stored(Z, K, phased);
    if (t < nst) break;
    t -= nst;
    rbase += bgen_values(Z, K, phased);
    ++i;
  }
  if (phased) {
    const int j = t / K, k = t % K;
    const long long b = rbase + (long long)j * (K - 1);
    if (k < K - 1) return bgen_value(packed, b + k, bits, scale);
    float remain = 1.0f;
    for (int kk = 0; kk < K - 1; ++kk) remain = bgen_sub(remain, bgen_value(packed, b + kk, bits, scale));
    return remain;
  }
  if (t < nst - 1) return bgen_value(packed, rbase + t, bits, scale);
  float remain = 1.0f;
  for (int kk = 0; kk < nst - 1; ++kk) remain = bgen_sub(remain, bgen_value(packed, rbase + kk, bits, scale));
  return remain;
}

// pass 1.  seg_count[variant * max_seg + segment] = packed values of the segment's samples
#if !defined(RVT_K_SPLIT) || defined(RVT_K_STREAM)
static __global__ __launch_bounds__(kBgenSeg) void bgen_count_kernel(const unsigned char* __restrict__ data,
                                                              const BgenRecord* __restrict__ rec, long long N,
                                                              int max_seg, long long* __restrict__ seg_count,
                                                              int* __restrict__ err) {
  const BgenRecord r = rec[blockIdx.y];
  if (r.layout != 2) return;
  const long long i = (long long)blockIdx.x * kBgenSeg + threadIdx.x;
  const unsigned char* pm = data + r.off + 8;
  int Z = (i < N) ? (pm[i] & 0x3f) : 0;
  if (Z > r.zmax) {  // (the host sized its checks by the declared maximum)
    atomicCAS_system(err, 0, (int)blockIdx.y + 1);
    Z = r.zmax;
  }
  long long n = (i < N) ? bgen_values(Z, r.K, r.phased) : 0;
  for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
  __shared__ long long ws[kBgenSeg / 64];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long s = 0;
    for (int w = 0; w < kBgenSeg / 64; ++w) s += ws[w];
    seg_count[(long long)blockIdx.y * max_seg + blockIdx.x] = s;
  }
}
#endif  // RVT_K_STREAM

// pass 2: exclusive scan over the segments of one variant (in place); a block whose packed area is shorter than its
// ploidy bytes demand raises *err (host-visible) to variant index + 1
#if !defined(RVT_K_SPLIT) || defined(RVT_K_STREAM)
static __global__ __launch_bounds__(256) void bgen_scan_kernel(const BgenRecord* __restrict__ rec, long long N, int max_seg,
                                                        long long* __restrict__ seg_count, int* __restrict__ err) {
  const BgenRecord r = rec[blockIdx.x];
  if (r.layout != 2) return;
  const int n_seg = (int)((N + kBgenSeg - 1) / kBgenSeg);
  long long* sc = seg_count + (long long)blockIdx.x * max_seg;
  __shared__ long long ws[4];
  __shared__ long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n_seg; base += 256) {
    const int i = base + threadIdx.x;
    const long long v = (i < n_seg) ? sc[i] : 0;
    long long x = v;
    for (int o = 1; o < 64; o <<= 1) {
      const long long y = __shfl_up(x, o);
      if ((threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = x;
    __syncthreads();
    long long before = carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += ws[w];
    if (i < n_seg) sc[i] = before + x - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = before + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const long long have = (r.len - 10 - N) * 8;
    if (carry * r.bits > have) atomicCAS_system(err, 0, (int)blockIdx.x + 1);
  }
}
#endif  // RVT_K_STREAM

// pass 3.  out[row_of_sample[i] + variant * ld] = the genotype of file sample i (rows the map never addresses are
// filled beforehand)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_STREAM)
static __global__ __launch_bounds__(kBgenSeg) void bgen_decode_kernel(const unsigned char* __restrict__ data,
                                                               const BgenRecord* __restrict__ rec, long long N,
                                                               int max_seg, const long long* __restrict__ seg_count,
                                                               const int* __restrict__ row_of_sample, long long ld,
                                                               double* __restrict__ out) {
  const BgenRecord r = rec[blockIdx.y];
  const long long i = (long long)blockIdx.x * kBgenSeg + threadIdx.x;
  const unsigned char* blk = data + r.off;
  double g = kBgenMissing;
  if (r.layout == 1) {
    if (i < N) {
      const unsigned short* v = reinterpret_cast<const unsigned short*>(blk + 6 * i);  // (6 i is even; blk 4-byte aligned)
      const float p0 = __fdiv_rn((float)v[0], 32768.0f), p1 = __fdiv_rn((float)v[1], 32768.0f),
                  p2 = __fdiv_rn((float)v[2], 32768.0f);
      if (!(p0 == 0 && p1 == 0 && p2 == 0)) g = __dadd_rn((double)p1, __dmul_rn((double)p2, 2.0));
    }
  } else {
    const unsigned char* pm = blk + 8;
    const unsigned char* packed = blk + 8 + N + 2;
    // (packed is read through 4-byte words: the host places the block so that blk + 8 + N + 2 is 4-byte aligned)
    const int byte = (i < N) ? pm[i] : 0;
    const int Z = byte & 0x3f;
    const long long nv = (i < N) ? bgen_values(Z, r.K, r.phased) : 0;
    long long x = nv;  // inclusive scan over the segment's threads
    for (int o = 1; o < 64; o <<= 1) {
      const long long y = __shfl_up(x, o);
      if ((threadIdx.x & 63) >= o) x += y;
    }
    __shared__ long long ws[kBgenSeg / 64];
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = x;
    __syncthreads();
    long long before = seg_count[(long long)blockIdx.y * max_seg + blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += ws[w];
    const long long rbase = before + x - nv;
    if (i < N && !(byte & 0x80) && (Z == 2 || Z == 1)) {
      if (r.K == 2) {
        const float s1 = bgen_stored_prob(pm, packed, N, r.K, r.phased, r.bits, r.scale, i, rbase, 1);
        const float s2 = bgen_stored_prob(pm, packed, N, r.K, r.phased, r.bits, r.scale, i, rbase, 2);
        g = __dadd_rn((double)s1, __dmul_rn((double)s2, 2.0));
      } else if (r.K == 1) {
        g = 2.0;
      } else {
        const float s0 = bgen_stored_prob(pm, packed, N, r.K, r.phased, r.bits, r.scale, i, rbase, 0);
        const float s1 = bgen_stored_prob(pm, packed, N, r.K, r.phased, r.bits, r.scale, i, rbase, 1);
        const float s2 = bgen_stored_prob(pm, packed, N, r.K, r.phased, r.bits, r.scale, i, rbase, 2);
        const double total = (double)bgen_add(bgen_add(s0, s1), s2);
        g = (total > 0.) ? __ddiv_rn(__dadd_rn((double)s1, __dmul_rn((double)s2, 2.0)), total) : kBgenMissing;
      }
    }
  }
  if (r.alt > 1) g = kBgenMissing;
  if (i < N) {
    const int row = row_of_sample ? row_of_sample[i] : (int)i;
    if (row >= 0) out[(long long)row + (long long)blockIdx.y * ld] = g;
  }
}
#endif  // RVT_K_STREAM

}  // namespace rvt
