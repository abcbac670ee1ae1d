// rvtests_amd — sufficient statistics of HARD-CALL genotype blocks against a WEIGHTED (binary-trait) null model.
//
// Same outputs as gene_suffstat_mfma<*, weighted> (suffstat_kernels.hip.h): R = G'V[G | X | rr] with V = diag(v),
// v_i = p_i (1 - p_i) of the logistic null model, exact column statistics, the burden sums of the collapsed genotypes.
// The fp64 kernel is bound by the fp64 matrix pipe there (16 instructions of 64 cycles per tile per 64 samples).  For
// hard calls the weighted Gram matrix sum_i v_i g_i g_i' goes to the int8 pipe as well:
//   * once per null model v_i (<= 1/4) is rounded to 42 fractional bits and split into SIX balanced base-128 digits
//     d_ip in [-64, 63], v_i = sum_p d_ip 128^-(p+1) (rvt_set_null; absolute error <= 2^-43 per weight, unbiased — the
//     size of the rounding the fp64 sums of 10^5 terms carry); stored per group of four samples as [plane 0..7][4 bytes],
//     so a lane's digits for the four samples of a step are 32 contiguous bytes at the same offset its genotype doubles
//     have inside a column;
//   * per 64-sample operand and plane the A operand is the byte-wise product d_p (x) g (|.| <= 128: a byte select among
//     0, d, 2 d — one v_perm_b32, see HcwRow), the B operand the packed genotypes: ONE v_mfma_i32_16x16x64_i8 (16 cycles) per plane and tile,
//     6 per tile against 16 fp64 instructions of 64 cycles;
//   * the six plane products of a tile are combined exactly in int32 (HcwAcc below);
//   * G'V[X | rr] stays on the fp64 matrix cores against the null tile [vX_0 .. vX_{d-1} | res | v | 0] (res = v rr);
//     its v column also gives the weighted burden sums c'Vc;
//   * column sums and the counts behind min / max are byte sums of g and g^2.
// Structure (load ring, raw buffer loads with range checks, wave-parts, burden collapse with predicted flips) as in
// suffstat_hc.hip.h, whose helpers it uses.
#pragma once
#include "suffstat_hc.hip.h"

namespace rvt {

constexpr int kHcwPlanes = 6;
constexpr int kHcwPairs = 3;
constexpr int kHcwMaxSteps = 3072;  // 16-sample steps per wave-part: keeps the int32 pair tiles exact (see above)
constexpr int kHcwMaxMT = 5;   // widest weighted hard-call class (M <= 80)
constexpr int kHcwMaxD = 13;   // X columns + res + v share ONE 16-column tile

struct NullTileW {
  const double* base;       // [vX_0 .. vX_{d-1} | res | v | zeros], ld doubles per column
  int cols;                 // d + 3
  const unsigned char* vq;  // digit planes of 2 v: 8 bytes per sample (32 per 4-sample group: [plane][4]), ld * 8 bytes
};

template <int MT>
struct HcwStep {
  u4_t glo[MT], ghi[MT];
  u4_t xlo, xhi;
  u4_t dq0, dq1;  // digits of planes 0-3 / 4-7 for the lane's four samples
};

template <int MT>
__device__ __forceinline__ void hcw_issue(HcwStep<MT>& f, const __amdgpu_buffer_rsrc_t& rg, const unsigned (&voff)[MT],
                                          const __amdgpu_buffer_rsrc_t& rx, unsigned xoff, const __amdgpu_buffer_rsrc_t& rq,
                                          unsigned qoff, int imm) {
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    f.glo[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c] + imm, 0, 0));
    f.ghi[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c] + imm + 16, 0, 0));
  }
  f.xlo = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + imm, 0, 0));
  f.xhi = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + imm + 16, 0, 0));
  f.dq0 = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rq, qoff + imm, 0, 0));
  f.dq1 = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rq, qoff + imm + 16, 0, 0));
}

struct HcwBurden {
  double a_cmc, a_zeg, a_zz;  // sum c x, sum c_zeg x, sum c_zeg^2 x over this lane's null column
  unsigned cnt;               // #(c != 0) over the samples of this lane's row
};

// one tile row of one step: fp64 tile of G'V[X | rr | v], packing, byte sums of g and g^2, burden hits
template <bool MASKED>
__device__ __forceinline__ void hcw_row(const u4_t& glo, const u4_t& ghi, const double (&xv)[4], d4_t& accT, unsigned& pk,
                                        unsigned& cs, unsigned& cs2, unsigned fx, unsigned& h, bool valid, unsigned& notHard) {
  // entries that are not hard calls (mk) void the gene here: the weighted kernel has no masked-entry tiles, the engine
  // runs such a gene on the fp64 kernel (wflags bit 1).  -inf (code 3, see hc_row) likewise.
  unsigned mk;
  hc_row<MASKED>(glo, ghi, xv, accT, pk, mk, cs, fx, h, valid);
  notHard |= mk | hc_code3(pk);
  cs2 = __builtin_amdgcn_sad_u8((pk & 0x01010101u) | ((pk & 0x02020202u) << 1), 0u, cs2);  // g^2: 0 / 1 / 4
}

// end of a step: the per-sample variant counts of this lane's row and the burden sums
template <bool MASKED>
__device__ __forceinline__ void hcw_finish(unsigned h, const double (&xv)[4], HcwBurden& bu, unsigned vmask) {
  h = row16_sum(h);
  if (MASKED) h &= vmask;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const unsigned cz = (h >> (8 * l)) & 0xffu;
    const unsigned cc = cz ? 1u : 0u;
    bu.cnt += cc;
    bu.a_zeg = fma((double)cz, xv[l], bu.a_zeg);
    bu.a_zz = fma((double)(cz * cz), xv[l], bu.a_zz);
    bu.a_cmc = fma((double)cc, xv[l], bu.a_cmc);
  }
}

// One step from a step buffer.  T = position of the step in its group of 4.
template <int MT, bool MASKED>
__device__ __forceinline__ void hcw_step(const HcwStep<MT>& f, const int T, d4_t (&accT)[MT], unsigned (&pk)[MT][4],
                                         unsigned (&dg)[4][8], unsigned (&cs)[MT], unsigned (&cs2)[MT],
                                         const unsigned (&fx)[MT], HcwBurden& bu, bool valid, unsigned vmask,
                                         unsigned& notHard) {
  double xv[4] = {hc_dbl(f.xlo[0], f.xlo[1]), hc_dbl(f.xlo[2], f.xlo[3]), hc_dbl(f.xhi[0], f.xhi[1]),
                  hc_dbl(f.xhi[2], f.xhi[3])};
  if (MASKED) {
#pragma unroll
    for (int l = 0; l < 4; ++l) xv[l] = valid ? xv[l] : 0.0;
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    dg[T][p] = (MASKED && !valid) ? 0u : f.dq0[p];
    dg[T][4 + p] = (MASKED && !valid) ? 0u : f.dq1[p];
  }
  unsigned h = 0;
#pragma unroll
  for (int c = 0; c < MT; ++c)
    hcw_row<MASKED>(f.glo[c], f.ghi[c], xv, accT[c], pk[c][T], cs[c], cs2[c], fx[c], h, valid, notHard);
  hcw_finish<MASKED>(h, xv, bu, vmask);
}

__device__ __forceinline__ double hcw_pow2(int e) { return __builtin_bit_cast(double, (unsigned long long)(1023 + e) << 52); }

// A operands of one row tile: byte-wise d g from the digits d and their doubles d2.  The product is a byte SELECT — 0, d or
// 2 d for g = 0, 1, 2 — so one v_perm_b32 per (plane, step) forms it from {d2, d} with a selector built once per step of
// the row: byte k of the selector is 0x0c (constant 0), k (byte k of d) or 4 + k (byte k of d2).  The selector itself
// comes from two table look-ups with the packed genotypes AS selector (v_perm_b32 again) — 3 instructions per step and
// row, 1 per plane and step, where the mask form ((d & m1) | (d2 & m2)) took 4 + 3.
template <int MT>
struct HcwRow {
  unsigned sel[4];
  __device__ __forceinline__ explicit HcwRow(const unsigned (&pkr)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const unsigned base = __builtin_amdgcn_perm(0u, 0x0004000cu, pkr[s]);  // g = 0 / 1 / 2 -> 0x0c / 0x00 / 0x04
      const unsigned nz = __builtin_amdgcn_perm(0u, 0x00ffff00u, pkr[s]);    // g = 0 / 1 / 2 -> 0x00 / 0xff / 0xff
      sel[s] = base | (nz & 0x03020100u);
    }
  }
  __device__ __forceinline__ i4_t aop(const unsigned (&dg)[4][8], const unsigned (&d2)[4][kHcwPlanes], int p) const {
    unsigned w[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) w[s] = __builtin_amdgcn_perm(d2[s][p], dg[s][p], sel[s]);
    return i4_t{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
  }
};

__device__ __forceinline__ void hcw_doubles(const unsigned (&dg)[4][8], unsigned (&d2)[4][kHcwPlanes]) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int p = 0; p < kHcwPlanes; ++p) d2[s][p] = (dg[s][p] << 1) & 0xFEFEFEFEu;  // 2 d per byte (|d| <= 64)
}

// The weighted Gram tiles sum_i v_i g_ri g_ci.  Planes are accumulated in PAIRS in exact int32 tiles that live across the
// whole wave-part: the even plane's fresh tile is shifted left by 7 bits and added to the pair's accumulator (one
// v_lshl_add_u32 per element), which is the C operand of the odd plane's instruction.  |pair sum| <= 2^21 + 2^14 per
// operand, so a wave-part may hold up to 1016 operands (the host cuts at kHcwMaxSteps = 3072 steps); the three pair tiles
// are combined in fp64 once, at the end.  12 vector instructions per tile and operand.  (Folding every plane, or chained
// triples of planes, into fp64 tiles per operand was measured slower in every class: the conversions cost more than
// the shifts.)
// Classes of three and five tiles keep the tiles of their LAST pairs (the low planes) in LDS instead of registers: 4 T
// registers less per lane and pair.  The workgroup is one wave, its LDS operations execute in order, and a tile costs one
// ds_read_b128 + one ds_write_b128 per 64-sample operand.  MT = 5: 120 registers less, no spilling (with all three pairs
// in registers the compiler spills 80 registers of the streaming loop to scratch memory: 4.2 instead of 5.4 TB/s
// isolated).  MT = 3: 48 registers less bring the kernel under 256, i.e. TWO waves per SIMD (5.2 -> 6.0 TB/s on the widths
// M = 33..48 of a batch; with one wave per SIMD the loop is bound by its own instruction issue, not by memory).  MT = 4
// does not get under 256 registers this way (20 KB of tiles per wave x 8 waves is also all of the CU's LDS).
#ifndef RVT_HCW_LDS_PAIRS_MT5
#define RVT_HCW_LDS_PAIRS_MT5 2
#endif
#ifndef RVT_HCW_LDS_PAIRS_MT4
#define RVT_HCW_LDS_PAIRS_MT4 0
#endif
#ifndef RVT_HCW_LDS_PAIRS_MT3
#define RVT_HCW_LDS_PAIRS_MT3 2
#endif
template <int MT>
constexpr int hcw_lds_pairs() {
  return MT >= 5 ? RVT_HCW_LDS_PAIRS_MT5 : (MT == 4 ? RVT_HCW_LDS_PAIRS_MT4 : (MT == 3 ? RVT_HCW_LDS_PAIRS_MT3 : 0));
}
template <int MT>
constexpr int hcw_lds_tiles() {
  return hcw_lds_pairs<MT>() * (MT * (MT + 1) / 2);
}

template <int MT>
struct HcwAcc {
  static constexpr int T = MT * (MT + 1) / 2;
  static constexpr int NL = hcw_lds_pairs<MT>();  // pairs kept in LDS: pair j >= R at lp[((j - R) T + t) 64 + lane]
  static constexpr int R = kHcwPairs - NL;         // pairs kept in registers
  i4_t p[R > 0 ? R : 1][T];
  __device__ __forceinline__ void init(i4_t* lp, int lane) {
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
      for (int t = 0; t < T; ++t) p[j][t] = i4_t{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NL * T; ++t) lp[t * 64 + lane] = i4_t{0, 0, 0, 0};
  }
  __device__ __forceinline__ double value(int t, int i, const i4_t* lp, int lane) const {  // pair j: weight 128^-(2j+2)
    double x = 0.0;
#pragma unroll
    for (int j = kHcwPairs - 1; j >= R; --j) x = fma((double)lp[((j - R) * T + t) * 64 + lane][i], hcw_pow2(-14 * (j + 1)), x);
#pragma unroll
    for (int j = R - 1; j >= 0; --j) x = fma((double)p[j][t][i], hcw_pow2(-14 * (j + 1)), x);
    return x;
  }
  __device__ __forceinline__ void gram(const unsigned (&pk)[MT][4], const unsigned (&dg)[4][8], i4_t* lp, int lane) {
    i4_t op[MT];
#pragma unroll
    for (int c = 0; c < MT; ++c) op[c] = i4_t{(int)pk[c][0], (int)pk[c][1], (int)pk[c][2], (int)pk[c][3]};
    unsigned d2[4][kHcwPlanes];
    hcw_doubles(dg, d2);
    int t0 = 0;
#pragma unroll
    for (int r = 0; r < MT; ++r) {
      const HcwRow<MT> row(pk[r]);
#pragma unroll
      for (int j = 0; j < kHcwPairs; ++j) {
        const i4_t a0 = row.aop(dg, d2, 2 * j), a1 = row.aop(dg, d2, 2 * j + 1);
        i4_t z[MT];
#pragma unroll
        for (int c = r; c < MT; ++c) z[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, op[c], i4_t{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
        for (int c = r; c < MT; ++c) {
          const bool in_lds = j >= R;
          i4_t* const lt = lp + ((in_lds ? j - R : 0) * T + t0 + c - r) * 64 + lane;
          i4_t acc = in_lds ? *lt : p[in_lds ? 0 : j][t0 + c - r];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = (int)(((unsigned)z[c][i] << 7) + (unsigned)acc[i]);
          acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, op[c], acc, 0, 0, 0);
          if (in_lds)
            *lt = acc;
          else
            p[in_lds ? 0 : j][t0 + c - r] = acc;
        }
      }
      t0 += MT - r;
      __builtin_amdgcn_sched_barrier(0);  // (row by row: keeps the live set to one row's operands)
    }
  }
};

template <int MT, int DEPTH>
__device__ __forceinline__ void suffstat_hcw_body(const GeneDesc& gd, const NullTileW& nt, long long N, long long ld,
                                                  int d, i4_t* lp) {
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
  auto uniform = [](const void* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
  };
  const unsigned gbytes = (unsigned)((unsigned long long)M * (unsigned long long)ld * 8ull);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(uniform(gd.G), 0, gbytes, 0x00020000);
  const unsigned xbytes = (unsigned)((unsigned long long)nt.cols * (unsigned long long)ld * 8ull);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(uniform(nt.base), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rq =
      __builtin_amdgcn_make_buffer_rsrc(uniform(nt.vq), 0, (unsigned)((unsigned long long)ld * 8ull), 0x00020000);
  const unsigned lane_off = (unsigned)(q * 32);
  const unsigned col_bytes = (unsigned)((unsigned long long)ld * 8ull);
  unsigned vbase[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    const int col = c * 16 + v;
    vbase[c] = (col < M) ? (unsigned)col * col_bytes + lane_off : 0x80000000u;
  }
  const int xcol = (v <= d + 1) ? v : d + 2;  // vX_k, res, v, or the zero column
  const unsigned xbase = (unsigned)xcol * col_bytes + lane_off;
  unsigned fx[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c) fx[c] = ((gd.pflip[c] >> v) & 1) ? 0x02020202u : 0u;

  d4_t accT[MT];
  HcwAcc<MT> acc;
  unsigned cs[MT], cs2[MT], pk[MT][4], dg[4][8];
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    accT[c] = d4_t{0.0, 0.0, 0.0, 0.0};
    cs[c] = cs2[c] = 0;
  }
  acc.init(lp, lane);
  HcwBurden bu{0.0, 0.0, 0.0, 0u};
  unsigned notHard = 0u;

  long long s = s_begin;
  const long long full = N >> 4;
  const long long s_fast_end = (s_end < full) ? s_end : full;
  constexpr int U = (DEPTH == 3) ? 12 : 4;  // (the host cuts the sample axis in multiples of kHcStepUnit = 12 steps)
  const long long n_fast = (s_fast_end > s_begin) ? (s_fast_end - s_begin) / U : 0;
  if constexpr (DEPTH == 1) {
   if (n_fast > 0) {
    // Rolling refill (as suffstat_hc_body, DEPTH = 1): ONE genotype step buffer; as soon as a tile row of step s has
    // been consumed its registers are the destination of the same row of step s + 1.  The null-model tile and the
    // digits are double-buffered.
    unsigned voff[MT];
#pragma unroll
    for (int c = 0; c < MT; ++c) voff[c] = vbase[c] + (unsigned)(s_begin * 128);
    unsigned xoff = xbase + (unsigned)(s_begin * 128);
    unsigned qoff = lane_off + (unsigned)(s_begin * 128);
    u4_t glo[MT], ghi[MT], xlo[2], xhi[2], dq0[2], dq1[2];
#pragma unroll
    for (int c = 0; c < MT; ++c) {
      glo[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c], 0, 0));
      ghi[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c] + 16, 0, 0));
    }
    xlo[0] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff, 0, 0));
    xhi[0] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + 16, 0, 0));
    dq0[0] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rq, qoff, 0, 0));
    dq1[0] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rq, qoff + 16, 0, 0));
    for (long long it = 0; it < n_fast; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int nb = (u + 1) & 1, cb = u & 1;
        xlo[nb] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + (u + 1) * 128, 0, 0));
        xhi[nb] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + (u + 1) * 128 + 16, 0, 0));
        dq0[nb] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rq, qoff + (u + 1) * 128, 0, 0));
        dq1[nb] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rq, qoff + (u + 1) * 128 + 16, 0, 0));
        __builtin_amdgcn_sched_barrier(0);
        const double xv[4] = {hc_dbl(xlo[cb][0], xlo[cb][1]), hc_dbl(xlo[cb][2], xlo[cb][3]),
                              hc_dbl(xhi[cb][0], xhi[cb][1]), hc_dbl(xhi[cb][2], xhi[cb][3])};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          dg[u][p] = dq0[cb][p];
          dg[u][4 + p] = dq1[cb][p];
        }
        unsigned h = 0;
#pragma unroll
        for (int c = 0; c < MT; ++c) {
          hcw_row<false>(glo[c], ghi[c], xv, accT[c], pk[c][u], cs[c], cs2[c], fx[c], h, true, notHard);
          glo[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c] + (u + 1) * 128, 0, 0));
          ghi[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c] + (u + 1) * 128 + 16, 0, 0));
          __builtin_amdgcn_sched_barrier(0);
        }
        hcw_finish<false>(h, xv, bu, 0xffffffffu);
        if (u == 3) acc.gram(pk, dg, lp, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int c = 0; c < MT; ++c) voff[c] += 4 * 128;
      xoff += 4 * 128;
      qoff += 4 * 128;
    }
    s += n_fast * U;
   }
  } else if (n_fast > 0) {
    unsigned voff[MT];
#pragma unroll
    for (int c = 0; c < MT; ++c) voff[c] = vbase[c] + (unsigned)(s_begin * 128);
    unsigned xoff = xbase + (unsigned)(s_begin * 128);
    unsigned qoff = lane_off + (unsigned)(s_begin * 128);
    HcwStep<MT> f[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH - 1; ++u) hcw_issue<MT>(f[u], rg, voff, rx, xoff, rq, qoff, u * 128);
    for (long long it = 0; it < n_fast; ++it) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        hcw_issue<MT>(f[(u + DEPTH - 1) % DEPTH], rg, voff, rx, xoff, rq, qoff, (u + DEPTH - 1) * 128);
        __builtin_amdgcn_sched_barrier(0);
        hcw_step<MT, false>(f[u % DEPTH], u & 3, accT, pk, dg, cs, cs2, fx, bu, true, 0xffffffffu, notHard);
        if ((u & 3) == 3) acc.gram(pk, dg, lp, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int c = 0; c < MT; ++c) voff[c] += U * 128;
      xoff += U * 128;
      qoff += U * 128;
    }
    s += n_fast * U;
  }
  while (s < s_end) {  // remainder: groups of 4 steps, each loaded from a clamped position and masked
#pragma unroll
    for (int c = 0; c < MT; ++c) pk[c][0] = pk[c][1] = pk[c][2] = pk[c][3] = 0u;
    auto one = [&](const int T, long long su) {
      const bool valid = su < s_end;
      const long long sc = valid ? su : s_end - 1;
      unsigned voff[MT];
#pragma unroll
      for (int c = 0; c < MT; ++c) voff[c] = vbase[c] + (unsigned)(sc * 128);
      HcwStep<MT> f;
      hcw_issue<MT>(f, rg, voff, rx, xbase + (unsigned)(sc * 128), rq, lane_off + (unsigned)(sc * 128), 0);
      unsigned vmask = 0u;
      const long long smp = sc * 16 + q * 4;
#pragma unroll
      for (int l = 0; l < 4; ++l) vmask |= (valid && smp + l < N) ? (0xffu << (8 * l)) : 0u;
      hcw_step<MT, true>(f, T, accT, pk, dg, cs, cs2, fx, bu, valid, vmask, notHard);
    };
    one(0, s);
    one(1, s + 1);
    one(2, s + 2);
    one(3, s + 3);
    acc.gram(pk, dg, lp, lane);
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
          for (int i = 0; i < 4; ++i) out[(long long)(r * 16 + q * 4 + i) * Cp + col] = acc.value(t, i, lp, lane);  // i32 map
        }
      }
  }
#pragma unroll
  for (int r = 0; r < MT; ++r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r * 16 + q + 4 * i;  // f64 C/D map
      if (M + v < Cp) out[(long long)row * Cp + M + v] = (v <= d) ? accT[r][i] : 0.0;  // (the v column is not part of R)
      if (M + 16 + v < Cp) out[(long long)row * Cp + M + 16 + v] = 0.0;
    }
  }
  // ---- column sum / min / max from the byte sums of g and g^2 ------------------------------------------------------
  long long cnt_w = ((s_end * 16 < N) ? s_end * 16 : N) - s_begin * 16;
  if (cnt_w < 0) cnt_w = 0;
  double* cst = gd.colstat + (long long)wpart * kHcColstatRows * gd.Mp;
  if (gd.wflags) {
    const bool nh = __builtin_amdgcn_ballot_w64(notHard != 0u) != 0ull;
    if (lane == 0) gd.wflags[wpart] = nh ? 2u : 0u;
  }
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    unsigned sc = cs[c], sq = cs2[c];
    sc += __shfl_xor(sc, 16, 64);
    sq += __shfl_xor(sq, 16, 64);
    sc += __shfl_xor(sc, 32, 64);
    sq += __shfl_xor(sq, 32, 64);
    const long long sm = (long long)sc, n2 = ((long long)sq - sm) / 2, n1 = 2 * sm - (long long)sq, n0 = cnt_w - n1 - n2;
    const double mn = n0 > 0 ? 0.0 : (n1 > 0 ? 1.0 : (n2 > 0 ? 2.0 : INFINITY));
    const double mx = n2 > 0 ? 2.0 : (n1 > 0 ? 1.0 : (n0 > 0 ? 0.0 : -INFINITY));
    if (lane < 16) {
      cst[c * 16 + lane] = (double)sm;
      cst[gd.Mp + c * 16 + lane] = mn;
      cst[2 * gd.Mp + c * 16 + lane] = mx;
      cst[3 * gd.Mp + c * 16 + lane] = 0.0;  // no masked entries on this path (rows 3-5 as suffstat_hc.hip.h writes them)
      reinterpret_cast<unsigned long long*>(cst)[4 * gd.Mp + c * 16 + lane] = 0ull;
      reinterpret_cast<unsigned long long*>(cst)[5 * gd.Mp + c * 16 + lane] = ~0ull;
    }
  }
  // ---- burden partial sums: [test][U, c'Vc, count, c'VX_0 .. c'VX_{d-1}], test 0 = CMC, 1 = Zeggini ------------------
  if (gd.bparts) {
    double ac = bu.a_cmc, az = bu.a_zeg, azz = bu.a_zz;
    ac += __shfl_xor(ac, 16, 64);
    az += __shfl_xor(az, 16, 64);
    azz += __shfl_xor(azz, 16, 64);
    ac += __shfl_xor(ac, 32, 64);
    az += __shfl_xor(az, 32, 64);
    azz += __shfl_xor(azz, 32, 64);
    unsigned cn = bu.cnt;
    cn += __shfl_xor(cn, 16, 64);
    cn += __shfl_xor(cn, 32, 64);
    const int rl = 3 + d;
    double* bp = gd.bparts + (long long)wpart * 2 * rl;
    if (lane <= d) {  // lane k < d: column v X_k; lane d: res
      const int k = (lane == d) ? 0 : 3 + lane;
      bp[k] = ac;
      bp[rl + k] = az;
    }
    if (lane == d + 1) {  // the v column: c'Vc (CMC: c^2 = c)
      bp[1] = ac;
      bp[rl + 1] = azz;
    }
    if (lane == 0) {
      bp[2] = (double)cn;
      bp[rl + 2] = (double)cn;
    }
  }
}

template <int MT, int DEPTH, int WAVES>
__global__ __launch_bounds__(64, WAVES) void gene_suffstat_hcw(const GeneDesc* __restrict__ genes, NullTileW nt,
                                                               long long N, long long ld, int d) {
  const GeneDesc gd = genes[blockIdx.y];
  if (gd.MT != MT) return;
  __shared__ i4_t lp[hcw_lds_tiles<MT>() > 0 ? hcw_lds_tiles<MT>() * 64 : 1];
  suffstat_hcw_body<MT, DEPTH>(gd, nt, N, ld, d, lp);
}

}  // namespace rvt
