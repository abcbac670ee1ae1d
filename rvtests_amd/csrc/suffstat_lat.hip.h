// rvtests_amd — sufficient statistics of DOSAGE blocks whose entries lie on a decimal lattice, against an unweighted
// (quantitative-trait) null model.
//
// `rvtest --dosage DS` reads the dosage of every sample from a VCF field printed with a fixed number of decimals
// (imputation servers write three: "0.998"); strtod turns it into the double nearest to K / den, K an integer, den =
// 10^decimals (src/VCFGenotypeExtractor / DataConsolidator hand those doubles to fit() unchanged).  For such a block G'G
// is, up to the factor 1 / den^2, an INTEGER matrix, and the kernel computes it exactly on the int8 matrix cores — the
// fp64 kernel (suffstat_kernels.hip.h) is bound by the fp64 matrix pipe at 0.45 of the HBM rate for these blocks, and no
// scheduling changes that (16 instructions of 64 cycles per tile and 64 samples):
//   * every loaded double g is mapped to K = rint(g den) with the magic-number addition t = fma(g, den, 1.5 2^52) — the
//     low dword of t IS K — and TESTED: |g den - K| <= K 2^-53 (one more fma gives the exact residual; the double nearest
//     to K / den is K / den (1 + delta) with |delta| <= 2^-53, so exactly the doubles strtod makes of the printed field
//     pass — a value a few ulps beside a lattice point does not, and 0 only as 0.0) and 0 <= g < 2 + 2^-19
//     (high dword <= 0x40000000; catches negative values, NaN and infinities).  A block that fails (arbitrary doubles:
//     BGEN's float probabilities, mean-imputed entries) is handed back to the fp64 kernel by gene_flags_hc_kernel, like
//     a dosage block that was sent to the hard-call kernel;
//   * K = L + 128 H with 7-bit digits (den <= kLatMaxDen = 2048: H <= 32): four steps of 16 samples are packed into the
//     int8 operands L_c and H_c of every 16-column tile c, and a tile of K'K is
//         L_r'L_c  +  128 (L_r'H_c + H_r'L_c)  +  16384 H_r'H_c
//     = FOUR v_mfma_i32_16x16x64_i8 (16 cycles each) against 16 fp64 instructions of 64 cycles.  Two int32 tiles per
//     tile of G'G live across the wave-part: `lo` = sum L'L, `hi` = sum [128 H'H + L'H + H'L] (the H'H product of an
//     operand is shifted left by 7 bits and added: one v_lshl_add_u32 per element).  All digits are non-negative, so the
//     tiles are exact as UNSIGNED 32-bit numbers: per operand lo <= 127^2 64 < 2^20, hi <= 128 32^2 64 + 2 127 32 64 <
//     2^23.1, and a wave-part holds at most kHcMaxSteps / 4 = 255 operands: < 2^31.1.  The partial tiles carry the integers
//     lo + 128 hi (exact in fp64, and so is their sum over the wave-parts); gene_assemble divides by den^2 ONCE per
//     element — the integer part is exact, where the fp64 kernel rounds every product and every sum;
//   * G'[X | rr] stays on the fp64 matrix cores with the loaded doubles (4 instructions per tile row and step), the
//     column sums are byte sums of the digits (integers, divided by den once in gene_assemble), min / max are packed 16-bit
//     min / max of K;
//   * the burden tests are computed in the same pass as in suffstat_hc.hip.h: a variant counts for a sample when
//     (int)g' > 0, i.e. g >= 1.0 for an unflipped column and g <= 1.0 for a flipped one — compared on the doubles
//     themselves, with the flips predicted from the caller's allele frequencies and verified by gene_flags_hc_kernel.
// Structure (wave-parts, load ring / rolling refill, raw buffer loads with range checks, partial-tile layout, the
// six-row column statistics, the flag word per wave-part) as in suffstat_hc.hip.h, whose helpers it uses.  No masked
// entries: a dosage block with mean-imputed entries (one off-lattice value per column) fails the test and takes the
// fp64 kernel — imputed data is complete.
#pragma once
#include "suffstat_hc.hip.h"

namespace rvt {

constexpr int kLatMaxMT = 5;       // widest lattice class (M <= 80)
constexpr int kLatMaxDen = 2048;   // K <= 2 den + 1 <= 4097: high digit <= 32 (the range proof above)
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));

struct LatParam {
  double den;   // lattice denominator (an integer, 1 .. kLatMaxDen)
};

// Tile classes whose `hi` tiles live in LDS instead of registers (4 T registers less; one wave per workgroup, so the LDS
// operations execute in order: one ds_read_b128 + one ds_write_b128 per tile and 64-sample operand).
#ifndef RVT_LAT_LDS_MT
#define RVT_LAT_LDS_MT 4
#endif
template <int MT>
constexpr int lat_lds_tiles() {
  return MT >= RVT_LAT_LDS_MT ? MT * (MT + 1) / 2 : 0;
}

struct LatCol {        // per lane and tile row: statistics of the lane's column
  unsigned s0, s1;     // byte sums of the low / high digits
  us2_t mn, mx;        // packed 16-bit min / max of K
};

// Per lane and tile row: what decides whether a value counts in the burden collapse.  (int)g' > 0 means g >= 1.0 for an
// unflipped column and g <= 1.0, i.e. NOT g > 1.0, for a flipped one.  For 0 <= g < 4 the doubles order like their bit
// patterns, and bits + C reaches bit 62 exactly when g >= 1.0 (C = 2^52) resp. g > 1.0 (C = 2^52 - 1): one 64-bit
// addition per value, bit 30 of the high dword, xor with the flip.
struct LatFlip {
  unsigned long long C;  // 0x0010000000000000 or 0x000FFFFFFFFFFFFF
  unsigned F;            // 0 or 0x01010101
};

// one tile row of one step: fp64 MFMAs for G'[X | rr], lattice test, digits, column statistics, burden hits
template <bool MASKED>
__device__ __forceinline__ void lat_row(u4_t glo, u4_t ghi, const double (&xv)[4], d4_t& accT, unsigned& b0,
                                        unsigned& b1, LatCol& st, const LatFlip& fl, unsigned& h, double& emax,
                                        unsigned& hmax, double den, bool valid, unsigned inv01, unsigned inv23) {
  if (MASKED && !valid) {
    glo = u4_t{0u, 0u, 0u, 0u};
    ghi = u4_t{0u, 0u, 0u, 0u};
  }
  const unsigned lo[4] = {glo[0], glo[2], ghi[0], ghi[2]}, hi[4] = {glo[1], glo[3], ghi[1], ghi[3]};
  constexpr double kMagic = 6755399441055744.0;  // 1.5 * 2^52
  unsigned K[4], sh[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const double g = hc_dbl(lo[l], hi[l]);
    accT = __builtin_amdgcn_mfma_f64_16x16x4f64(g, xv[l], accT, 0, 0, 0);
    const double t = __builtin_fma(g, den, kMagic);
    const double r = t - kMagic;
    const double e = __builtin_fma(g, den, -r);
    emax = __builtin_fmax(emax, __builtin_fma(r, -0x1p-53, __builtin_fabs(e)));  // |g den - K| - K 2^-53: must stay <= 0
    K[l] = (unsigned)__builtin_bit_cast(unsigned long long, t);
    sh[l] = (unsigned)(((((unsigned long long)hi[l]) << 32 | lo[l]) + fl.C) >> 32);
  }
  {
    const unsigned a = hi[0] > hi[1] ? hi[0] : hi[1], b = hi[2] > hi[3] ? hi[2] : hi[3];
    const unsigned m = a > b ? a : b;
    hmax = hmax > m ? hmax : m;
  }
  // top bytes of the four sums -> one dword (selectors as hc_row), bit 6 of every byte = the comparison
  const unsigned top = __builtin_amdgcn_perm(sh[1], sh[0], 0x0c0c0703u) | __builtin_amdgcn_perm(sh[3], sh[2], 0x07030c0cu);
  h += ((top >> 6) & 0x01010101u) ^ fl.F;  // (hc_finish masks the samples that do not exist)
  const unsigned w01 = K[0] | (K[1] << 16), w23 = K[2] | (K[3] << 16);
  b0 = __builtin_amdgcn_perm(w23, w01, 0x06040200u) & 0x7F7F7F7Fu;
  b1 = __builtin_amdgcn_perm(w23 >> 7, w01 >> 7, 0x06040200u) & 0x7F7F7F7Fu;
  st.s0 = __builtin_amdgcn_sad_u8(b0, 0u, st.s0);
  st.s1 = __builtin_amdgcn_sad_u8(b1, 0u, st.s1);
  if (MASKED) {  // samples beyond N (or a step beyond the wave's range) take no part in min / max
    st.mn = __builtin_elementwise_min(st.mn, __builtin_bit_cast(us2_t, w01 | inv01));
    st.mn = __builtin_elementwise_min(st.mn, __builtin_bit_cast(us2_t, w23 | inv23));
    st.mx = __builtin_elementwise_max(st.mx, __builtin_bit_cast(us2_t, w01 & ~inv01));
    st.mx = __builtin_elementwise_max(st.mx, __builtin_bit_cast(us2_t, w23 & ~inv23));
  } else {
    st.mn = __builtin_elementwise_min(st.mn, __builtin_bit_cast(us2_t, w01));
    st.mn = __builtin_elementwise_min(st.mn, __builtin_bit_cast(us2_t, w23));
    st.mx = __builtin_elementwise_max(st.mx, __builtin_bit_cast(us2_t, w01));
    st.mx = __builtin_elementwise_max(st.mx, __builtin_bit_cast(us2_t, w23));
  }
  // Everything this row produces is consumed only at the end of the group of four steps; without this the compiler sinks
  // the whole computation down there and keeps the loaded doubles of all four steps alive (3 x the registers).
  unsigned mn = __builtin_bit_cast(unsigned, st.mn), mx = __builtin_bit_cast(unsigned, st.mx);
  asm volatile("" : "+v"(b0), "+v"(b1), "+v"(st.s0), "+v"(st.s1), "+v"(mn), "+v"(mx), "+v"(h), "+v"(emax), "+v"(hmax));
  st.mn = __builtin_bit_cast(us2_t, mn);
  st.mx = __builtin_bit_cast(us2_t, mx);
}

// the two int32 tile sets of a wave-part
template <int MT>
struct LatAcc {
  static constexpr int T = MT * (MT + 1) / 2;
  static constexpr bool LDS_HI = lat_lds_tiles<MT>() > 0;
  i4_t lo[T];
  i4_t hi[LDS_HI ? 1 : T];
  __device__ __forceinline__ void init(i4_t* lp, int lane) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      lo[t] = i4_t{0, 0, 0, 0};
      if (LDS_HI)
        lp[t * 64 + lane] = i4_t{0, 0, 0, 0};
      else
        hi[t] = i4_t{0, 0, 0, 0};
    }
  }
  __device__ __forceinline__ double value(int t, int i, const i4_t* lp, int lane) const {  // the integer (K'K)_ij
    const unsigned vh = LDS_HI ? (unsigned)lp[t * 64 + lane][i] : (unsigned)hi[LDS_HI ? 0 : t][i];
    return (double)(unsigned)lo[t][i] + 128.0 * (double)vh;  // < 2^39: exact
  }
  __device__ __forceinline__ void gram(const unsigned (&b0)[MT][4], const unsigned (&b1)[MT][4], i4_t* lp, int lane) {
    i4_t L[MT], H[MT];
#pragma unroll
    for (int c = 0; c < MT; ++c) {
      L[c] = i4_t{(int)b0[c][0], (int)b0[c][1], (int)b0[c][2], (int)b0[c][3]};
      H[c] = i4_t{(int)b1[c][0], (int)b1[c][1], (int)b1[c][2], (int)b1[c][3]};
    }
    int t = 0;
#pragma unroll
    for (int r = 0; r < MT; ++r) {
      i4_t z[MT];
#pragma unroll
      for (int c = r; c < MT; ++c) z[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(H[r], H[c], i4_t{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
      for (int c = r; c < MT; ++c) lo[t + c - r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(L[r], L[c], lo[t + c - r], 0, 0, 0);
#pragma unroll
      for (int c = r; c < MT; ++c) {
        i4_t* const lt = lp + (t + c - r) * 64 + lane;
        i4_t acc = LDS_HI ? *lt : hi[LDS_HI ? 0 : t + c - r];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (int)(((unsigned)z[c][i] << 7) + (unsigned)acc[i]);
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(L[r], H[c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(H[r], L[c], acc, 0, 0, 0);
        if (LDS_HI)
          *lt = acc;
        else
          hi[LDS_HI ? 0 : t + c - r] = acc;
      }
      t += MT - r;
      __builtin_amdgcn_sched_barrier(0);  // (row by row: keeps the live set to one row's fresh tiles)
    }
  }
};

template <int MT>
struct LatState {
  d4_t accT[MT];
  LatCol col[MT];
  unsigned b0[MT][4], b1[MT][4];
  LatFlip fl[MT];
  HcBurden bu;
  double emax;
  unsigned hmax;
};

// One step from a step buffer.  T = position of the step in its group of 4.
template <int MT, bool MASKED>
__device__ __forceinline__ void lat_step(const HcStep<MT>& f, const int T, LatState<MT>& S, double den, bool valid,
                                         unsigned vmask) {
  double xv[4] = {hc_dbl(f.xlo[0], f.xlo[1]), hc_dbl(f.xlo[2], f.xlo[3]), hc_dbl(f.xhi[0], f.xhi[1]),
                  hc_dbl(f.xhi[2], f.xhi[3])};
  unsigned inv01 = 0u, inv23 = 0u;
  if (MASKED) {
#pragma unroll
    for (int l = 0; l < 4; ++l) xv[l] = valid ? xv[l] : 0.0;
    inv01 = ((vmask & 0xffu) ? 0u : 0xffffu) | ((vmask & 0xff00u) ? 0u : 0xffff0000u);
    inv23 = ((vmask & 0xff0000u) ? 0u : 0xffffu) | ((vmask & 0xff000000u) ? 0u : 0xffff0000u);
  }
  unsigned h = 0;
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    lat_row<MASKED>(f.glo[c], f.ghi[c], xv, S.accT[c], S.b0[c][T], S.b1[c][T], S.col[c], S.fl[c], h, S.emax, S.hmax, den, valid,
                    inv01, inv23);
    __builtin_amdgcn_sched_barrier(0);  // (row by row: the fp64 temporaries of a row die before the next one starts)
  }
  hc_finish<MASKED>(h, xv, S.bu, vmask);
}

template <int MT, int DEPTH>
__device__ __forceinline__ void suffstat_lat_body(const GeneDesc& gd, const NullTile& nt, const LatParam lp_, long long N,
                                                  long long ld, int d, i4_t* lp) {
  const int lane = threadIdx.x & 63;
  const int v = lane & 15, q = lane >> 4;
  const int wpart = blockIdx.x;
  if (wpart >= gd.n_wparts) return;
  const long long nsteps = ld >> 4;
  const long long s_begin = (long long)wpart * gd.steps_per_wpart;
  long long s_end = s_begin + gd.steps_per_wpart;
  if (s_end > nsteps) s_end = nsteps;
  if (s_begin >= s_end) return;
  const int M = gd.M;
  const double den = lp_.den;
  auto uniform = [](const void* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
  };
  const unsigned gbytes = (unsigned)((unsigned long long)M * (unsigned long long)ld * 8ull);  // < 2^31 (host checks)
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(uniform(gd.G), 0, gbytes, 0x00020000);
  const unsigned xbytes = (unsigned)((unsigned long long)nt.cols * (unsigned long long)ld * 8ull);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(uniform(nt.base), 0, xbytes, 0x00020000);
  const unsigned lane_off = (unsigned)(q * 32);
  const unsigned col_bytes = (unsigned)((unsigned long long)ld * 8ull);
  unsigned vbase[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    const int col = c * 16 + v;
    vbase[c] = (col < M) ? (unsigned)col * col_bytes + lane_off : 0x80000000u;
  }
  const int xcol = (v <= d) ? v : d + 1;  // X_k, rr, or the zero column
  const unsigned xbase = (unsigned)xcol * col_bytes + lane_off;

  LatState<MT> S;
  LatAcc<MT> acc;
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    S.accT[c] = d4_t{0.0, 0.0, 0.0, 0.0};
    S.col[c].s0 = S.col[c].s1 = 0u;
    S.col[c].mn = us2_t{0xffff, 0xffff};
    S.col[c].mx = us2_t{0, 0};
    const bool flipped = ((gd.pflip[c] >> v) & 1) != 0;
    S.fl[c].C = flipped ? 0x000FFFFFFFFFFFFFull : 0x0010000000000000ull;
    S.fl[c].F = flipped ? 0x01010101u : 0u;
  }
  S.bu = HcBurden{0.0, 0.0, 0u, 0u};
  S.emax = 0.0;
  S.hmax = 0u;
  acc.init(lp, lane);

  long long s = s_begin;
  const long long full = N >> 4;  // steps whose 16 samples all exist
  const long long s_fast_end = (s_end < full) ? s_end : full;
  constexpr int U = (DEPTH == 3) ? 12 : 4;
  const long long n_fast = (s_fast_end > s_begin) ? (s_fast_end - s_begin) / U : 0;
  if constexpr (DEPTH == 1) {
   if (n_fast > 0) {
    // rolling refill (suffstat_hc_body, DEPTH = 1): one genotype step buffer, refilled row by row
    unsigned voff[MT];
#pragma unroll
    for (int c = 0; c < MT; ++c) voff[c] = vbase[c] + (unsigned)(s_begin * 128);
    unsigned xoff = xbase + (unsigned)(s_begin * 128);
    u4_t glo[MT], ghi[MT], xlo[2], xhi[2];
#pragma unroll
    for (int c = 0; c < MT; ++c) {
      glo[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c], 0, 0));
      ghi[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c] + 16, 0, 0));
    }
    xlo[0] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff, 0, 0));
    xhi[0] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + 16, 0, 0));
    for (long long it = 0; it < n_fast; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xlo[(u + 1) & 1] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + (u + 1) * 128, 0, 0));
        xhi[(u + 1) & 1] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + (u + 1) * 128 + 16, 0, 0));
        __builtin_amdgcn_sched_barrier(0);
        const double xv[4] = {hc_dbl(xlo[u & 1][0], xlo[u & 1][1]), hc_dbl(xlo[u & 1][2], xlo[u & 1][3]),
                              hc_dbl(xhi[u & 1][0], xhi[u & 1][1]), hc_dbl(xhi[u & 1][2], xhi[u & 1][3])};
        unsigned h = 0;
#pragma unroll
        for (int c = 0; c < MT; ++c) {
          lat_row<false>(glo[c], ghi[c], xv, S.accT[c], S.b0[c][u], S.b1[c][u], S.col[c], S.fl[c], h, S.emax, S.hmax, den,
                         true, 0u, 0u);
          glo[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c] + (u + 1) * 128, 0, 0));
          ghi[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c] + (u + 1) * 128 + 16, 0, 0));
          __builtin_amdgcn_sched_barrier(0);
        }
        hc_finish<false>(h, xv, S.bu, 0xffffffffu);
        if (u == 3) acc.gram(S.b0, S.b1, lp, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int c = 0; c < MT; ++c) voff[c] += 4 * 128;
      xoff += 4 * 128;
    }
    s += n_fast * U;
   }
  } else if (n_fast > 0) {
    unsigned voff[MT];
#pragma unroll
    for (int c = 0; c < MT; ++c) voff[c] = vbase[c] + (unsigned)(s_begin * 128);
    unsigned xoff = xbase + (unsigned)(s_begin * 128);
    HcStep<MT> f[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH - 1; ++u) hc_issue<MT, false>(f[u], rg, voff, rx, xoff, u * 128);
    for (long long it = 0; it < n_fast; ++it) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        hc_issue<MT, false>(f[(u + DEPTH - 1) % DEPTH], rg, voff, rx, xoff, (u + DEPTH - 1) * 128);
        __builtin_amdgcn_sched_barrier(0);
        lat_step<MT, false>(f[u % DEPTH], u & 3, S, den, true, 0xffffffffu);
        if ((u & 3) == 3) acc.gram(S.b0, S.b1, lp, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int c = 0; c < MT; ++c) voff[c] += U * 128;
      xoff += U * 128;
    }
    s += n_fast * U;
  }
  // ---- remainder: groups of 4 steps, each step loaded from a clamped position and masked ------------------------
  while (s < s_end) {
#pragma unroll
    for (int c = 0; c < MT; ++c)
#pragma unroll
      for (int u = 0; u < 4; ++u) S.b0[c][u] = S.b1[c][u] = 0u;
    auto one = [&](const int T, long long su) {
      const bool valid = su < s_end;
      const long long sc = valid ? su : s_end - 1;
      unsigned voff[MT];
#pragma unroll
      for (int c = 0; c < MT; ++c) voff[c] = vbase[c] + (unsigned)(sc * 128);
      HcStep<MT> f;
      hc_issue<MT, false>(f, rg, voff, rx, xbase + (unsigned)(sc * 128), 0);
      unsigned vmask = 0u;
      const long long smp = sc * 16 + q * 4;
#pragma unroll
      for (int l = 0; l < 4; ++l) vmask |= (valid && smp + l < N) ? (0xffu << (8 * l)) : 0u;
      lat_step<MT, true>(f, T, S, den, valid, vmask);
    };
    one(0, s);
    one(1, s + 1);
    one(2, s + 2);
    one(3, s + 3);
    acc.gram(S.b0, S.b1, lp, lane);
    s += 4;
  }

  // ---- partial tiles: element (row, col) -> parts[row * Cp + col], the layout gene_assemble reduces ---------------
  double* out = gd.parts + (long long)wpart * gd.Mp * gd.Cp;
  const int Cp = gd.Cp;
  {
    int t = 0;
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
      for (int c = r; c < MT; ++c, ++t) {
        const int col = c * 16 + v;
        if (col < M) {
#pragma unroll
          for (int i = 0; i < 4; ++i) out[(long long)(r * 16 + q * 4 + i) * Cp + col] = acc.value(t, i, lp, lane);
        }
      }
  }
#pragma unroll
  for (int r = 0; r < MT; ++r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r * 16 + q + 4 * i;  // f64 C/D map
      if (M + v < Cp) out[(long long)row * Cp + M + v] = S.accT[r][i];
      if (M + 16 + v < Cp) out[(long long)row * Cp + M + 16 + v] = 0.0;
    }
  }
  // ---- the wave-part's flag: bit 1 = a value off the lattice (or outside [0, 2]): the gene goes to the fp64 kernel ----
  {
    const bool bad = !(S.emax <= 0.0) || S.hmax > 0x40000000u;
    const bool any = __builtin_amdgcn_ballot_w64(bad) != 0ull;
    if (gd.wflags && lane == 0) gd.wflags[wpart] = any ? 2u : 0u;
  }
  // ---- column statistics (rows as suffstat_hc.hip.h writes them; no masked entries on this path) -----------------------
  double* cst = gd.colstat + (long long)wpart * kHcColstatRows * gd.Mp;
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    unsigned s0 = S.col[c].s0, s1 = S.col[c].s1;
    unsigned mn = S.col[c].mn[0] < S.col[c].mn[1] ? S.col[c].mn[0] : S.col[c].mn[1];
    unsigned mx = S.col[c].mx[0] > S.col[c].mx[1] ? S.col[c].mx[0] : S.col[c].mx[1];
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      s0 += __shfl_xor(s0, o, 64);
      s1 += __shfl_xor(s1, o, 64);
      const unsigned a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
      mn = a < mn ? a : mn;
      mx = b > mx ? b : mx;
    }
    if (lane < 16) {
      const int j = c * 16 + lane;
      const bool none = mn > mx;  // no sample in this wave-part
      cst[j] = (double)((unsigned long long)s0 + 128ull * s1);  // the integer sum of K (divided once, after the reduction)
      cst[gd.Mp + j] = none ? INFINITY : (double)mn / den;
      cst[2 * gd.Mp + j] = none ? -INFINITY : (double)mx / den;
      cst[3 * gd.Mp + j] = 0.0;
      reinterpret_cast<unsigned long long*>(cst)[4 * gd.Mp + j] = 0ull;
      reinterpret_cast<unsigned long long*>(cst)[5 * gd.Mp + j] = ~0ull;
    }
  }
  // ---- burden partial sums: [test][U, c'c, count, c'X_0 .. c'X_{d-1}], test 0 = CMC, 1 = Zeggini ---------------------
  if (gd.bparts) {
    double ac = S.bu.a_cmc, az = S.bu.a_zeg;
    ac += __shfl_xor(ac, 16, 64);
    az += __shfl_xor(az, 16, 64);
    ac += __shfl_xor(ac, 32, 64);
    az += __shfl_xor(az, 32, 64);
    unsigned zz = S.bu.zz, cn = S.bu.cnt;
    zz += __shfl_xor(zz, 16, 64);
    cn += __shfl_xor(cn, 16, 64);
    zz += __shfl_xor(zz, 32, 64);
    cn += __shfl_xor(cn, 32, 64);
    const int rl = 3 + d;
    double* bp = gd.bparts + (long long)wpart * 2 * rl;
    if (lane <= d) {
      const int k = (lane == d) ? 0 : 3 + lane;
      bp[k] = ac;
      bp[rl + k] = az;
    }
    if (lane == 0) {
      bp[1] = (double)cn;
      bp[2] = (double)cn;
      bp[rl + 1] = (double)zz;
      bp[rl + 2] = (double)cn;
    }
  }
}

template <int MT, int DEPTH, int WAVES>
__global__ __launch_bounds__(64, WAVES) void gene_suffstat_lat(const GeneDesc* __restrict__ genes, NullTile nt,
                                                               LatParam lp_, long long N, long long ld, int d) {
  __shared__ i4_t lp[lat_lds_tiles<MT>() > 0 ? lat_lds_tiles<MT>() * 64 : 1];
  const GeneDesc gd = genes[blockIdx.y];
  if (gd.MT != MT) return;
  suffstat_lat_body<MT, DEPTH>(gd, nt, lp_, N, ld, d, lp);
}

}  // namespace rvt
