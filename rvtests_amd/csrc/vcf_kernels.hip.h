// rvtests_amd — VCF text -> hard-call genotype bytes on the device (SURVEY §8f "next" #1).
//
// Replaces the per-sample loop of VCFGenotypeExtractor::extractMultipleGenotype (src/VCFGenotypeExtractor.cpp:29-140):
// for every record the reference splits the sample columns at tabs (libVcf/VCFRecord.h), every column at ':'
// (VCFIndividual::parse, libVcf/VCFIndividual.h:27-58), takes the subfield at the FORMAT index of "GT"
// (VCFIndividual::justGet, :88-93: an index past the last subfield yields the default value "."), decodes it with
// VCFValue::getGenotype (libVcf/VCFValue.h:74-117) and turns it into MISSING_GENOTYPE (-9, libVcf/VCFConstant.h:4) when
// the GD / GQ depth and quality filters reject it (src/VCFGenotypeExtractor.cpp:304-317,397-439).
//
// The caller hands over the raw text of the sample columns of each record (everything after the FORMAT column, no
// newline); the host only reads the first nine columns (rvt_vcf_locate).  On the device a record is cut into 4 KiB
// segments; tabs are counted per segment (SWAR byte compare on 16-byte loads), an exclusive scan over the segments
// gives every tab its ordinal = the index of the sample column that starts behind it, and the thread that owns the tab
// decodes that column.  The output is one signed byte per kept sample (0 / 1 / 2, missing = -9) in the row order the
// caller's sample map prescribes — exactly what the int8 entry point (consolidate_* kernels, rvt_engine.hip) consumes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rvt {

constexpr int kVcfSegBytes = 4096;  // 256 threads x 16 bytes
constexpr int kVcfMissing = -9;

struct VcfRecord {          // one record (variant) of a gene
  long long text_off;       // offset of its sample columns inside the text buffer (multiple of 16)
  long long len;            // bytes
  int gt_idx, gd_idx, gq_idx;  // FORMAT indices (-1: not present)
  int alt;                     // 0: bi-allelic coding (getGenotype); a > 0: count of alternative allele a (multi-allelic mode)
  int hemi;                    // 1: the record lies in a hemizygous region (ParRegion::isHemiRegion): males are recoded
};
struct VcfFilters {
  int gd_min, gd_max, gq_min, gq_max;  // <= 0: off (the reference's `GDmin > 0 &&` tests)
};

// 0x80 in every byte of v that equals the byte b (exact per byte, no borrow artefacts)
__device__ __forceinline__ unsigned vcf_eq_mask(unsigned v, unsigned b4) {
  const unsigned x = v ^ b4;
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}

// tab mask of the 16 bytes at `pos` of a record of `len` bytes: bit (8 i + 7) of m[k] <=> byte 4 k + i is a tab
__device__ __forceinline__ int vcf_tabs16(const char* __restrict__ text, long long pos, long long len, unsigned (&m)[4]) {
  int n = 0;
  if (pos >= len) {
    m[0] = m[1] = m[2] = m[3] = 0u;
    return 0;
  }
  const uint4 v = *reinterpret_cast<const uint4*>(text + pos);
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  const long long left = len - pos;  // valid bytes from pos
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned t = vcf_eq_mask(w[k], 0x09090909u);
    const long long vb = left - 4 * k;
    if (vb < 4) t &= (vb <= 0) ? 0u : ((1u << (8 * (int)vb)) - 1u);
    m[k] = t;
    n += __popc(t);
  }
  return n;
}

// pass 1: tabs per segment.  grid (max segments, records), 256 threads
#if !defined(RVT_K_SPLIT) || defined(RVT_K_STREAM)
static __global__ __launch_bounds__(256) void vcf_tab_count_kernel(const char* __restrict__ text,
                                                            const VcfRecord* __restrict__ rec, int max_seg,
                                                            int* __restrict__ seg_count) {
  const VcfRecord r = rec[blockIdx.y];
  const long long seg0 = (long long)blockIdx.x * kVcfSegBytes;
  if (seg0 >= r.len) return;
  unsigned m[4];
  int n = vcf_tabs16(text + r.text_off, seg0 + 16 * threadIdx.x, r.len, m);
  for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
  __shared__ int ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) seg_count[(long long)blockIdx.y * max_seg + blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
#endif  // RVT_K_STREAM

// pass 2: exclusive scan over the segments of one record (in place); a record whose column count differs from the
// file's sample count raises *err (host-visible) to record index + 1.  grid (records), 256 threads
#if !defined(RVT_K_SPLIT) || defined(RVT_K_STREAM)
static __global__ __launch_bounds__(256) void vcf_tab_scan_kernel(const VcfRecord* __restrict__ rec, int max_seg,
                                                           int n_file_samples, int* __restrict__ seg_count,
                                                           int* __restrict__ err) {
  const VcfRecord r = rec[blockIdx.x];
  const int n_seg = (int)((r.len + kVcfSegBytes - 1) / kVcfSegBytes);
  int* sc = seg_count + (long long)blockIdx.x * max_seg;
  __shared__ int ws[4];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n_seg; base += 256) {
    const int i = base + threadIdx.x;
    const int v = (i < n_seg) ? sc[i] : 0;
    int x = v;  // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if ((threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = x;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += ws[w];
    if (i < n_seg) sc[i] = before + x - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = before + x;
    __syncthreads();
  }
  if (threadIdx.x == 0 && carry != n_file_samples - 1) atomicCAS_system(err, 0, (int)blockIdx.x + 1);
}
#endif  // RVT_K_STREAM

// ---- one sample column -------------------------------------------------------------------------------------------------
// [b, e) of subfield idx of the column that starts at `start`; false when the column has fewer subfields (the
// reference then reads its default value ".")
__device__ __forceinline__ bool vcf_subfield(const char* __restrict__ t, long long start, long long len, int idx,
                                             long long* b, long long* e) {
  if (idx < 0) return false;
  long long p = start;
  for (int k = 0;; ++k) {
    long long q = p;
    while (q < len && t[q] != '\t' && t[q] != ':') ++q;
    if (k == idx) {
      *b = p;
      *e = q;
      return true;
    }
    if (q >= len || t[q] == '\t') return false;
    p = q + 1;
  }
}

// VCFValue::getGenotype (libVcf/VCFValue.h:74-117) on the bytes [b, e); `signed char` comparisons as on the host
__device__ __forceinline__ int vcf_gt_code(const char* __restrict__ t, long long b, long long e) {
  if (b >= e) return kVcfMissing;  // empty: the terminator '\0' is < '0'
  const signed char c0 = (signed char)t[b];
  if (c0 == '.' || c0 < '0') return kVcfMissing;
  int g = c0 - '0';
  if (g > 1) return kVcfMissing;  // multi-allelic
  if (b + 1 == e) return g;       // haploid call
  const signed char c1 = (signed char)t[b + 1];
  if (c1 != '|' && c1 != '/') return kVcfMissing;
  if (b + 2 == e) return kVcfMissing;
  const signed char c2 = (signed char)t[b + 2];
  if (c2 == '.') return kVcfMissing;
  if (!(c2 < '0')) {  // (a byte below '0' is only reported by the reference; the first allele stands)
    const int a2 = c2 - '0';
    if (a2 > 1) return kVcfMissing;
    g += a2;
  }
  if (b + 3 != e) return kVcfMissing;
  return g;
}

// VCFValue::countAltAllele(alt) (libVcf/VCFValue.h:180-213; multi-allelic mode, src/VCFGenotypeExtractor.cpp:441-484): single
// digit allele indices; an allele that is not a digit is only reported and counts as "not alt"
__device__ __forceinline__ int vcf_alt_code(const char* __restrict__ t, long long b, long long e, int alt) {
  if (b >= e) return kVcfMissing;  // (empty subfield: malformed for the reference too)
  const signed char c0 = (signed char)t[b];
  if (c0 == '.') return kVcfMissing;
  int g = (c0 - '0' == alt) ? 1 : 0;
  if (b + 1 == e) return g;
  const signed char c1 = (signed char)t[b + 1];
  if (c1 != '|' && c1 != '/') return kVcfMissing;
  if (b + 2 == e) return kVcfMissing;
  const signed char c2 = (signed char)t[b + 2];
  if (c2 == '.') return kVcfMissing;
  if (!(c2 < '0' || c2 > '9')) g += (c2 - '0' == alt) ? 1 : 0;
  if (b + 3 != e) return kVcfMissing;
  return g;
}

// VCFValue::getAllele1 / getAllele2 (libVcf/VCFValue.h:159-179); a byte below '0' is only reported and reads as allele 0
__device__ __forceinline__ int vcf_allele(const char* __restrict__ t, long long p, long long e) {
  if (p >= e) return 0;  // (allele 1 of an empty subfield: the terminator is below '0')
  const signed char c = (signed char)t[p];
  if (c == '.') return kVcfMissing;
  return c < '0' ? 0 : c - '0';
}
// a MALE in a hemizygous region: VCFValue::getMaleNonParGenotype02 (libVcf/VCFValue.h:125-142; alt = 0) or
// countMaleNonParAltAllele2 (:214-236; alt > 0) — haploid calls count double, diploid calls must be homozygous
__device__ __forceinline__ int vcf_male_code(const char* __restrict__ t, long long b, long long e, int alt) {
  const int g = vcf_allele(t, b, e);
  if (g == kVcfMissing) return kVcfMissing;
  if (e - b == 1) {
    if (alt > 0) return g == alt ? 2 : 0;
    return g == 0 ? 0 : (g == 1 ? 2 : kVcfMissing);
  }
  if (b + 2 >= e) return kVcfMissing;
  const int g2 = vcf_allele(t, b + 2, e);
  if (g2 == kVcfMissing || g != g2) return kVcfMissing;
  if (alt > 0) return g == alt ? 2 : 0;
  return g == 0 ? 0 : (g == 1 ? 2 : kVcfMissing);
}

// atoi() of the subfield (NUL-terminated at its end in the reference)
__device__ __forceinline__ int vcf_atoi(const char* __restrict__ t, long long b, long long e) {
  while (b < e && (t[b] == ' ' || (t[b] >= '\t' && t[b] <= '\r'))) ++b;
  bool neg = false;
  if (b < e && (t[b] == '-' || t[b] == '+')) neg = t[b++] == '-';
  long long v = 0;
  while (b < e && t[b] >= '0' && t[b] <= '9' && v < (1LL << 40)) v = 10 * v + (t[b++] - '0');
  if (v > 2147483647LL) v = 2147483647LL;
  return neg ? -(int)v : (int)v;
}

// sex: PLINK code of the sample (1 male, 2 female, else unknown); only looked at in a hemizygous record
// (src/VCFGenotypeExtractor.cpp:416-426: males through the non-PAR rule, females as usual, unknown sex -> missing)
__device__ __forceinline__ int vcf_decode_column(const char* __restrict__ t, long long start, long long len,
                                                 const VcfRecord& r, const VcfFilters& f, int sex) {
  long long b, e;
  int g = kVcfMissing;
  if (r.hemi && sex != 2) {
    if (sex == 1 && r.gt_idx >= 0) {
      if (vcf_subfield(t, start, len, r.gt_idx, &b, &e))
        g = vcf_male_code(t, b, e, r.alt);
      // (an absent subfield reads the default value ".": missing)
    }
  } else if (vcf_subfield(t, start, len, r.gt_idx, &b, &e))  // else "." -> missing
    g = r.alt > 0 ? vcf_alt_code(t, b, e, r.alt) : vcf_gt_code(t, b, e);
  if (f.gd_min > 0 || f.gd_max > 0) {                                            // checkGD (:304-310)
    const int gd = vcf_subfield(t, start, len, r.gd_idx, &b, &e) ? vcf_atoi(t, b, e) : 0;  // atoi(".") = 0
    if ((f.gd_min > 0 && gd < f.gd_min) || (f.gd_max > 0 && gd > f.gd_max)) g = kVcfMissing;
  }
  if (f.gq_min > 0 || f.gq_max > 0) {                                            // checkGQ (:311-317)
    const int gq = vcf_subfield(t, start, len, r.gq_idx, &b, &e) ? vcf_atoi(t, b, e) : 0;
    if ((f.gq_min > 0 && gq < f.gq_min) || (f.gq_max > 0 && gq > f.gq_max)) g = kVcfMissing;
  }
  return g;
}

// atof() of a subfield (VCFValue::toDouble, libVcf/VCFValue.h:38-41) for the dosage mode: optional sign, decimal digits
// with an optional point, optional exponent.  With at most 15 significant digits and a decimal exponent within +-22 the
// value is ONE correctly rounded operation on exact doubles (digits * 10^e or digits / 10^-e) — exactly what a
// correctly rounding strtod returns.  Anything else (more digits, huge exponents, inf / nan / hex) raises *inexact;
// an empty or non-numeric field is 0.0, as atof gives.
__device__ __forceinline__ double vcf_atof(const char* __restrict__ t, long long b, long long e, int* inexact) {
  while (b < e && (t[b] == ' ' || (t[b] >= '\t' && t[b] <= '\r'))) ++b;
  bool neg = false;
  if (b < e && (t[b] == '-' || t[b] == '+')) neg = t[b++] == '-';
  unsigned long long mant = 0;
  int digits = 0, exp10 = 0;
  bool any = false, seen_nonzero = false;
  while (b < e && t[b] >= '0' && t[b] <= '9') {
    any = true;
    if (t[b] != '0' || seen_nonzero) {
      seen_nonzero = true;
      if (digits < 19) {
        mant = mant * 10 + (unsigned)(t[b] - '0');
        ++digits;
      } else {
        ++exp10;
        *inexact = 1;
      }
    }
    ++b;
  }
  if (b < e && t[b] == '.') {
    ++b;
    while (b < e && t[b] >= '0' && t[b] <= '9') {
      any = true;
      if (t[b] != '0' || seen_nonzero) {
        seen_nonzero = true;
        if (digits < 19) {
          mant = mant * 10 + (unsigned)(t[b] - '0');
          ++digits;
          --exp10;
        } else {
          *inexact = 1;
        }
      } else {
        --exp10;  // leading zeros behind the point
      }
      ++b;
    }
  }
  if (!any) {
    if (b < e && (t[b] == 'i' || t[b] == 'I' || t[b] == 'n' || t[b] == 'N')) *inexact = 1;  // inf / nan
    return 0.0;
  }
  if (b < e && (t[b] == 'x' || t[b] == 'X') && mant == 0) *inexact = 1;  // hex float
  if (b < e && (t[b] == 'e' || t[b] == 'E')) {
    long long q = b + 1;
    bool eneg = false;
    if (q < e && (t[q] == '-' || t[q] == '+')) eneg = t[q++] == '-';
    if (q < e && t[q] >= '0' && t[q] <= '9') {
      int ex = 0;
      while (q < e && t[q] >= '0' && t[q] <= '9' && ex < 10000) ex = ex * 10 + (t[q++] - '0');
      exp10 += eneg ? -ex : ex;
    }
  }
  if (mant == 0) return neg ? -0.0 : 0.0;
  if (digits > 15 || exp10 > 22 || exp10 < -22) *inexact = 1;
  const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                          1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  double v = (double)mant;
  const int ae = exp10 < 0 ? -exp10 : exp10;
  const double scale = p10[ae > 22 ? 22 : ae];
  v = exp10 < 0 ? __ddiv_rn(v, scale) : __dmul_rn(v, scale);
  return neg ? -v : v;
}

// dosage mode of the decode pass (--dosage TAG): the subfield at the tag's FORMAT index through atof; a column without
// that subfield reads the default value "." = 0.0; the GD / GQ filters turn a value into -9 as for hard calls.
// out: [record][ld] doubles, the first n_rows of every column pre-filled with -9.
#if !defined(RVT_K_SPLIT) || defined(RVT_K_STREAM)
static __global__ __launch_bounds__(256) void vcf_decode_dosage_kernel(const char* __restrict__ text,
                                                                const VcfRecord* __restrict__ rec, int max_seg,
                                                                const int* __restrict__ seg_count,
                                                                const int* __restrict__ row_of_sample,
                                                                const signed char* __restrict__ sex, int n_file_samples,
                                                                long long ld, VcfFilters flt, double* __restrict__ out,
                                                                int* __restrict__ err) {
  const VcfRecord r = rec[blockIdx.y];
  const long long seg0 = (long long)blockIdx.x * kVcfSegBytes;
  if (seg0 >= r.len) return;
  const char* t = text + r.text_off;
  double* col = out + (long long)blockIdx.y * ld;
  const long long pos = seg0 + 16 * threadIdx.x;
  unsigned m[4];
  const int n = vcf_tabs16(t, pos, r.len, m);
  int x = n;
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o);
    if ((threadIdx.x & 63) >= o) x += y;
  }
  __shared__ int ws[4];
  if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = x;
  __syncthreads();
  int k = seg_count[(long long)blockIdx.y * max_seg + blockIdx.x] + x - n;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) k += ws[w];
  int inexact = 0;
  auto column = [&](long long start, int s) {
    long long b, e;
    double g = vcf_subfield(t, start, r.len, r.gt_idx, &b, &e) ? vcf_atof(t, b, e, &inexact) : 0.0;
    if (r.hemi && sex && sex[s] == 1) g = g * 2.0;  // a male's dosage in a hemizygous region (exact)
    if (r.gt_idx < 0) g = (double)kVcfMissing;  // "Cannot find <tag> field!"
    if (flt.gd_min > 0 || flt.gd_max > 0) {
      const int gd = vcf_subfield(t, start, r.len, r.gd_idx, &b, &e) ? vcf_atoi(t, b, e) : 0;
      if ((flt.gd_min > 0 && gd < flt.gd_min) || (flt.gd_max > 0 && gd > flt.gd_max)) g = (double)kVcfMissing;
    }
    if (flt.gq_min > 0 || flt.gq_max > 0) {
      const int gq = vcf_subfield(t, start, r.len, r.gq_idx, &b, &e) ? vcf_atoi(t, b, e) : 0;
      if ((flt.gq_min > 0 && gq < flt.gq_min) || (flt.gq_max > 0 && gq > flt.gq_max)) g = (double)kVcfMissing;
    }
    return g;
  };
  if (blockIdx.x == 0 && threadIdx.x == 0 && n_file_samples > 0) {
    const int row = row_of_sample[0];
    if (row >= 0) col[row] = column(0, 0);
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    unsigned mm = m[w];
    while (mm) {
      const int bit = __ffs(mm) - 1;
      mm &= mm - 1;
      const int s = k + 1;
      ++k;
      if (s < n_file_samples) {
        const int row = row_of_sample[s];
        if (row >= 0) col[row] = column(pos + 4 * w + (bit >> 3) + 1, s);
      }
    }
  }
  if (inexact) atomicCAS_system(err, 0, -((int)blockIdx.y + 1));  // negative: a number the device cannot round exactly
}
#endif  // RVT_K_STREAM

// pass 3: decode.  grid (max segments, records), 256 threads.  out: [record][n_rows] signed bytes, pre-filled with -9
#if !defined(RVT_K_SPLIT) || defined(RVT_K_STREAM)
static __global__ __launch_bounds__(256) void vcf_decode_kernel(const char* __restrict__ text, const VcfRecord* __restrict__ rec,
                                                         int max_seg, const int* __restrict__ seg_count,
                                                         const int* __restrict__ row_of_sample,
                                                         const signed char* __restrict__ sex, int n_file_samples,
                                                         long long n_rows, VcfFilters flt, signed char* __restrict__ out) {
  const VcfRecord r = rec[blockIdx.y];
  const long long seg0 = (long long)blockIdx.x * kVcfSegBytes;
  if (seg0 >= r.len) return;
  const char* t = text + r.text_off;
  signed char* col = out + (long long)blockIdx.y * n_rows;
  const long long pos = seg0 + 16 * threadIdx.x;
  unsigned m[4];
  const int n = vcf_tabs16(t, pos, r.len, m);
  int x = n;  // exclusive prefix of the tab counts inside the block
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o);
    if ((threadIdx.x & 63) >= o) x += y;
  }
  __shared__ int ws[4];
  if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = x;
  __syncthreads();
  int k = seg_count[(long long)blockIdx.y * max_seg + blockIdx.x] + x - n;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) k += ws[w];
  if (blockIdx.x == 0 && threadIdx.x == 0 && n_file_samples > 0) {  // the first column has no tab in front of it
    const int row = row_of_sample[0];
    if (row >= 0) col[row] = (signed char)vcf_decode_column(t, 0, r.len, r, flt, sex ? sex[0] : 0);
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    unsigned mm = m[w];
    while (mm) {
      const int bit = __ffs(mm) - 1;  // 8 i + 7
      mm &= mm - 1;
      const int s = k + 1;  // the column behind the k-th tab (0-based) is sample k + 1
      ++k;
      if (s < n_file_samples) {
        const int row = row_of_sample[s];
        if (row >= 0)
          col[row] = (signed char)vcf_decode_column(t, pos + 4 * w + (bit >> 3) + 1, r.len, r, flt, sex ? sex[s] : 0);
      }
    }
  }
}
#endif  // RVT_K_STREAM

// dst[i + j * ld] = value for i < n_rows (the dosage matrix before the decode pass: every row missing)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_STREAM)
static __global__ void vcf_fill_kernel(double* __restrict__ dst, long long n_rows, long long ld, int ncols, double value) {
  const long long total = n_rows * ncols;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x)
    dst[(t / n_rows) * ld + t % n_rows] = value;
}
#endif  // RVT_K_STREAM

}  // namespace rvt
