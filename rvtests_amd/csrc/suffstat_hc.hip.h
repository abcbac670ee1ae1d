// rvtests_amd — sufficient statistics of HARD-CALL genotype blocks (every entry exactly 0.0, 1.0 or 2.0) against an
// unweighted (quantitative-trait) null model.  Same inputs, outputs and one-pass structure as gene_suffstat_mfma
// (suffstat_kernels.hip.h; replaces Skat.cpp:47-76, SkatO.cpp:124-182, LinearRegressionScoreTest.cpp:209-212,
// DataConsolidator.cpp:46-69,94-116 and the collapsers Model.cpp:73-89,115-130), with the arithmetic matched to what
// the data is:
//
//   * G'G is an INTEGER matrix.  The 8-byte doubles are streamed from HBM exactly as before (the boundary layout does
//     not change: 8 B per genotype, read once by one lane); each value is reduced to its 2-bit integer straight from
//     the top byte of the double (0.0 / 1.0 / 2.0 = 0x00.. / 0x3F.. / 0x40..), four steps are packed into the 16 int8
//     of a v_mfma_i32_16x16x64_i8 operand, and ONE such instruction per tile replaces 16 v_mfma_f64_16x16x4_f64
//     (64 cycles each).  The int32 accumulators are exact.  As in the fp64 kernel the register that holds 16 variants
//     x 64 samples is the A operand of a tile row and the B operand of a tile column.
//   * G'[X | rr] stays on the fp64 matrix cores with the loaded doubles as A operand and ONE shared tile of null-model
//     columns as B operand: MT x 4 instructions per step.
//   * column sums are byte sums (v_sad_u8) of the packed integers; min / max follow from the exact counts
//     n2 = (S_jj - s_j) / 2, n1 = 2 s_j - S_jj, n0 = n - n1 - n2.
//   * the burden tests are computed in the same pass: per sample the number of variants with (int)g' > 0 is a
//     packed-byte sum over the lane's tile rows followed by a 16-lane DPP reduction; whether a column is flipped
//     (g' = 2 - g) is PREDICTED from the allele frequency the caller supplies (af > 0.5) and verified afterwards against
//     the exact column sums (gene_flags_hc_kernel); a gene whose prediction was wrong, or that holds a monomorphic
//     non-zero column, gets its burden sums recomputed by burden_fallback_kernel.  The bit-mask planes and the
//     burden_collapse_kernel launch of the fp64 path are not needed.
//
//   * NOTHING is assumed about the block.  Every loaded double is tested (low dword zero, high dword one of the three
//     patterns) in the registers it already sits in.  An entry that is not a hard call is a MASKED entry m = 1 with
//     integer part H = 0: mean imputation (imputeGenotypeToMean, src/DataConsolidator.cpp:217-245) turns a column with
//     missing calls into H_j + mu_j m_j — integer H, 0/1 mask m, ONE value mu_j per column.  The kernel keeps such
//     columns on the integer pipe: the packed operand byte carries H + 4 m (0, 1, 2, 4), so the ordinary Gram tiles hold
//     C = (H + 4m)'(H + 4m); when a 64-sample operand holds a masked entry (wave-uniform test) two more sets of tiles,
//     P' = (H + 4m)'m and Q = m'm, are formed with the same instruction and added to 16-bit counters in LDS
//     (ds_add_u32 on packed pairs).  gene_assemble recovers H'H = C - 4(P + P') - 16 Q, P = P' - 4 Q, and
//     G'G = H'H + P diag(mu) + diag(mu) P' + diag(mu) Q diag(mu): exact integers, one rounding per product.
//     G'[X | rr] needs no correction (the fp64 operands are the true values).  That all masked entries of a column are
//     bit-identical is verified with LDS atomics (OR and AND of the masked bit patterns per column); a column where
//     they are not — dosages — or a value that is not finite sends the gene to the general fp64 kernel (flags[2 MT + 1],
//     gene_flags_hc_kernel; the engine re-runs it).  No separate classification pass, no trust in what a block held
//     when it was last looked at.
//
// Matrix pipe ~30 % busy, vector ALU ~25 %: the kernel is bound by HBM alone, which the fp64 kernel (matrix pipe
// co-limited at M ~ 50) is not.
#pragma once
#include <type_traits>
#include "suffstat_kernels.hip.h"

namespace rvt {

typedef int i4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

constexpr int kHcMaxMT = 6;      // widest hard-call class (M <= 96), as the single-pass fp64 kernel
constexpr int kHcMaxD = 13;      // the null-model columns + rr share ONE 16-column tile
constexpr int kHcStepUnit = 12;  // a wave's step count is a multiple of this (ring of 3 buffers x groups of 4 steps)
constexpr int kHcMaxSteps = 1020; // steps per wave-part: the 16-bit LDS counters of P' hold <= 4 x 16 x 1020 < 65536
constexpr int kHcColstatRows = 6; // sum of H, min, max (hard calls only), masked count, OR / AND of the masked bit patterns

// LDS of one wave (= one workgroup), 32-bit words:
//   [0, PQ)            P' tiles (MT x MT, row tile = H + 4m side) then Q tiles (upper triangle): [tile][half][lane],
//                      each word two 16-bit counters (accumulator elements 0|1 resp. 2|3 of the lane)
//   [PQ, PQ + 64 MT)   per column c * 16 + v: OR lo, OR hi, AND lo, AND hi of the masked entries' bit patterns
//   [PQ + 64 MT]       bit 0: the wave-part met a masked entry (the P'/Q image is written out), bit 1: an entry passed
//                      the hard-call test with code 3 (-inf): the gene goes to the general kernel
constexpr int hc_pq_tiles(int MT) { return MT * MT + MT * (MT + 1) / 2; }
constexpr int hc_pq_words(int MT) { return hc_pq_tiles(MT) * 128; }
constexpr int hc_lds_words(int MT) { return hc_pq_words(MT) + 64 * MT + 4; }

// Null-model tile of the hard-call kernel: ONE allocation [X_0 .. X_{d-1} | rr | zeros], ld doubles per column.
struct NullTile {
  const double* base;
  int cols;  // d + 2
};

// every lane ends with the sum over its 16-lane row (packed bytes cannot overflow: <= 6 tile rows x 16 lanes)
__device__ __forceinline__ unsigned row16_sum(unsigned h) {
  h += (unsigned)__builtin_amdgcn_update_dpp(0, (int)h, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  h += (unsigned)__builtin_amdgcn_update_dpp(0, (int)h, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  h += (unsigned)__builtin_amdgcn_update_dpp(0, (int)h, 0x141, 0xF, 0xF, true);  // row_half_mirror
  h += (unsigned)__builtin_amdgcn_update_dpp(0, (int)h, 0x140, 0xF, 0xF, true);  // row_mirror
  return h;
}

// one step of one lane: 4 consecutive samples (32 B = two 16-byte loads) of MT genotype columns and one null column
template <int MT>
struct HcStep {
  u4_t glo[MT], ghi[MT];
  u4_t xlo, xhi;
};

template <int MT, bool NT>
__device__ __forceinline__ void hc_issue(HcStep<MT>& f, const __amdgpu_buffer_rsrc_t& rg, const unsigned (&voff)[MT],
                                         const __amdgpu_buffer_rsrc_t& rx, unsigned xoff, int imm) {
  // aux bit 1 = nt: G is streamed once by exactly one CU; keep it from evicting the null-model columns
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    f.glo[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c] + imm, 0, NT ? 2 : 0));
    f.ghi[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c] + imm + 16, 0, NT ? 2 : 0));
  }
  f.xlo = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + imm, 0, 0));
  f.xhi = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + imm + 16, 0, 0));
}

__device__ __forceinline__ double hc_dbl(unsigned lo, unsigned hi) {
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

struct HcBurden {       // per-lane running sums of the collapsed-genotype score statistics
  double a_cmc, a_zeg;  // sum c * x over this lane's null column (x = X_k or rr)
  unsigned zz, cnt;     // sum c_zeg^2 and #(c != 0) over the samples of this lane's row
};

// one tile row of one step: fp64 MFMAs for G'[X | rr], hard-call test, packing, column sum, burden hits.
// pk = the integer genotypes H (0 / 1 / 2; 0 where masked), mk = 0x01 in the byte of every entry that is NOT exactly
// 0.0, 1.0 or 2.0.  A double is a hard call iff its low dword is zero and its high dword is 0, 0x3FF00000 or
// 0x40000000: hi + 0x00100000 then has no bit outside {20, 30}.  (0xFFF00000 = -inf passes too, with code 3: caught by
// the caller through hc_code3.)
template <bool MASKED>
__device__ __forceinline__ void hc_row(const u4_t& glo, const u4_t& ghi, const double (&xv)[4], d4_t& accT,
                                       unsigned& pk, unsigned& mk, unsigned& cs, unsigned fx, unsigned& h, bool valid) {
  const double g0 = hc_dbl(glo[0], glo[1]), g1 = hc_dbl(glo[2], glo[3]), g2 = hc_dbl(ghi[0], ghi[1]),
               g3 = hc_dbl(ghi[2], ghi[3]);
  accT = __builtin_amdgcn_mfma_f64_16x16x4f64(g0, xv[0], accT, 0, 0, 0);
  accT = __builtin_amdgcn_mfma_f64_16x16x4f64(g1, xv[1], accT, 0, 0, 0);
  accT = __builtin_amdgcn_mfma_f64_16x16x4f64(g2, xv[2], accT, 0, 0, 0);
  accT = __builtin_amdgcn_mfma_f64_16x16x4f64(g3, xv[3], accT, 0, 0, 0);
  // top bytes of the four doubles -> one dword (v_perm_b32: selector 0-3 = bytes of the 2nd operand, 4-7 = bytes of
  // the 1st, 0x0c = 0x00), then 0x00 / 0x3F / 0x40 -> 0 / 1 / 2 in every byte
  const unsigned w01 = __builtin_amdgcn_perm(glo[3], glo[1], 0x0c0c0703u);
  const unsigned w23 = __builtin_amdgcn_perm(ghi[3], ghi[1], 0x07030c0cu);
  unsigned p = ((w01 | w23) >> 5) & 0x03030303u;
  constexpr unsigned kAdd = 0x00100000u, kBits = 0xBFEFFFFFu;
  const unsigned i0 = ((glo[1] + kAdd) & kBits) | glo[0], i1 = ((glo[3] + kAdd) & kBits) | glo[2],
                 i2 = ((ghi[1] + kAdd) & kBits) | ghi[0], i3 = ((ghi[3] + kAdd) & kBits) | ghi[2];
  auto one = [](unsigned x) { return x < 1u ? x : 1u; };  // v_min_u32
  unsigned m = one(i0) | (one(i1) << 8) | (one(i2) << 16) | (one(i3) << 24);
  if (MASKED) {
    p = valid ? p : 0u;
    m = valid ? m : 0u;
  }
  const unsigned m3 = m * 3u;
  p &= ~m3;
  pk = p;
  mk = m;
  cs = __builtin_amdgcn_sad_u8(p, 0u, cs);
  const unsigned t = (p ^ fx) & ~m3;  // flipped column: (int)(2 - g) > 0  <=>  g != 2; a masked entry never counts here
  h += (t | (t >> 1)) & 0x01010101u;  // (gene_flags_hc_kernel sends the gene to the fallback when its mu says it should)
}
__device__ __forceinline__ unsigned hc_code3(unsigned p) { return p & (p >> 1) & 0x01010101u; }

// masked entries of one tile row of one step (rare path): OR / AND of their bit patterns per column, in LDS
__device__ __forceinline__ void hc_note_masked(const u4_t& glo, const u4_t& ghi, unsigned mk, unsigned* oa) {
  auto note = [&](unsigned lo, unsigned hi) {
    __hip_atomic_fetch_or(oa + 0, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_or(oa + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_and(oa + 2, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_and(oa + 3, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  if (mk & 0x000000ffu) note(glo[0], glo[1]);
  if (mk & 0x0000ff00u) note(glo[2], glo[3]);
  if (mk & 0x00ff0000u) note(ghi[0], ghi[1]);
  if (mk & 0xff000000u) note(ghi[2], ghi[3]);
}

// end of a step: the per-sample variant counts of this lane's row and the burden sums
template <bool MASKED>
__device__ __forceinline__ void hc_finish(unsigned h, const double (&xv)[4], HcBurden& bu, unsigned vmask) {
  h = row16_sum(h);  // byte l: variants with (int)g' > 0 for sample l of this lane's row
  if (MASKED) h &= vmask;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const unsigned cz = (h >> (8 * l)) & 0xffu;
    const unsigned cc = cz ? 1u : 0u;
    bu.zz = cz * cz + bu.zz;
    bu.cnt += cc;
    bu.a_zeg = fma((double)cz, xv[l], bu.a_zeg);
    bu.a_cmc = fma((double)cc, xv[l], bu.a_cmc);
  }
}

// per-wave state of the masked-entry bookkeeping: where this lane's columns keep their OR / AND words, the flag word
struct HcMaskCtx {
  unsigned* oa;    // LDS: OR lo, OR hi, AND lo, AND hi of column 0 * 16 + v; row tile c at oa + 64 c
  unsigned* flag;  // LDS flag word (see hc_lds_words)
};

// One step from a step buffer.  T = position of the step in its group of 4.  anym collects the mask bytes of the group.
template <int MT, bool MASKED>
__device__ __forceinline__ void hc_step(const HcStep<MT>& f, const int T, d4_t (&accT)[MT], unsigned (&pk)[MT][4],
                                        unsigned (&cs)[MT], const unsigned (&fx)[MT], HcBurden& bu, bool valid,
                                        unsigned vmask, unsigned& anym, const HcMaskCtx& mc) {
  double xv[4] = {hc_dbl(f.xlo[0], f.xlo[1]), hc_dbl(f.xlo[2], f.xlo[3]), hc_dbl(f.xhi[0], f.xhi[1]),
                  hc_dbl(f.xhi[2], f.xhi[3])};
  if (MASKED) {
#pragma unroll
    for (int l = 0; l < 4; ++l) xv[l] = valid ? xv[l] : 0.0;
  }
  unsigned h = 0;
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    unsigned p, m;
    hc_row<MASKED>(f.glo[c], f.ghi[c], xv, accT[c], p, m, cs[c], fx[c], h, valid);
    if (m) hc_note_masked(f.glo[c], f.ghi[c], m, mc.oa + 64 * c);
    if (hc_code3(p)) __hip_atomic_fetch_or(mc.flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    pk[c][T] = p | (m << 2);  // operand byte H + 4 m
    anym |= m;
  }
  hc_finish<MASKED>(h, xv, bu, vmask);
}

template <int MT>
__device__ __forceinline__ void hc_gram(const unsigned (&pk)[MT][4], i4_t (&accS)[MT * (MT + 1) / 2]) {
  i4_t op[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c) op[c] = i4_t{(int)pk[c][0], (int)pk[c][1], (int)pk[c][2], (int)pk[c][3]};
  int t = 0;
#pragma unroll
  for (int r = 0; r < MT; ++r)
#pragma unroll
    for (int c = r; c < MT; ++c, ++t) accS[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(op[r], op[c], accS[t], 0, 0, 0);
}

// The masked tiles of one 64-sample operand (only when it holds a masked entry): P' = (H + 4m)'m and Q = m'm, every
// element <= 256, added to the 16-bit LDS counters as packed pairs.
template <int MT>
__device__ __forceinline__ void hc_gram_masked(const unsigned (&pk)[MT][4], unsigned* lds, int lane) {
  // (the mask operands are rebuilt where they are used: 8 transient registers instead of 4 MT — this path runs beside a
  // full load ring)
  auto mask_op = [&](int c) {
    return i4_t{(int)((pk[c][0] >> 2) & 0x01010101u), (int)((pk[c][1] >> 2) & 0x01010101u),
                (int)((pk[c][2] >> 2) & 0x01010101u), (int)((pk[c][3] >> 2) & 0x01010101u)};
  };
  auto add = [&](int tile, const i4_t& z) {
    const unsigned w0 = (unsigned)z[0] | ((unsigned)z[1] << 16), w1 = (unsigned)z[2] | ((unsigned)z[3] << 16);
    __hip_atomic_fetch_add(lds + (tile * 2 + 0) * 64 + lane, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(lds + (tile * 2 + 1) * 64 + lane, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    const i4_t mc = mask_op(c);
#pragma unroll
    for (int r = 0; r < MT; ++r) {
      const i4_t a = i4_t{(int)pk[r][0], (int)pk[r][1], (int)pk[r][2], (int)pk[r][3]};
      add(r * MT + c, __builtin_amdgcn_mfma_i32_16x16x64_i8(a, mc, i4_t{0, 0, 0, 0}, 0, 0, 0));
      if (r <= c) {  // Q tile (r, c): index MT^2 + r MT - r (r - 1) / 2 + (c - r)
        const i4_t mr = (r == c) ? mc : mask_op(r);
        add(MT * MT + r * MT - r * (r - 1) / 2 + (c - r), __builtin_amdgcn_mfma_i32_16x16x64_i8(mr, mc, i4_t{0, 0, 0, 0}, 0, 0, 0));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __hip_atomic_fetch_or(lds + hc_pq_words(MT) + 64 * MT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// end of a group of 4 steps: the Gram tiles, and the masked tiles when any lane met a masked entry
template <int MT>
__device__ __forceinline__ void hc_group_end(const unsigned (&pk)[MT][4], i4_t (&accS)[MT * (MT + 1) / 2], unsigned& anym,
                                             unsigned* lds, int lane) {
  hc_gram<MT>(pk, accS);
  if (__builtin_amdgcn_ballot_w64(anym != 0u) != 0ull) hc_gram_masked<MT>(pk, lds, lane);
  anym = 0u;
}

// The wave's sample range [s_begin, s_end) (16-sample steps) is processed in iterations of U steps (groups of 4 = one
// int8 operand each) through a ring of DEPTH step buffers with DEPTH - 1 steps in flight (U = 12 for DEPTH = 3, else
// 4).  Whole iterations inside [0, N) run branch-free with immediate offsets; the remainder (only the last wave of a
// gene has one) is loaded step by step from clamped positions and masked.
template <int MT, int DEPTH, bool NT>
__device__ __forceinline__ void suffstat_hc_body(const GeneDesc& gd, const NullTile& nt, long long N, long long ld,
                                                 int d, unsigned* lds) {
  const int lane = threadIdx.x & 63;
  const int v = lane & 15, q = lane >> 4;
  const int wpart = blockIdx.x;
  if (wpart >= gd.n_wparts) return;
  // ---- LDS of this wave: masked-tile counters = 0, OR words = 0, AND words = ~0, flag = 0 (one wave per workgroup: LDS
  // operations of a wave execute in order, no barrier needed) ---------------------------------------------------------
  constexpr int kPQ = hc_pq_words(MT);
#pragma unroll 4
  for (int w = lane; w < kPQ; w += 64) lds[w] = 0u;
#pragma unroll
  for (int c = 0; c < MT; ++c) lds[kPQ + 64 * c + lane] = (lane & 2) ? 0xffffffffu : 0u;
  if (lane < 4) lds[kPQ + 64 * MT + lane] = 0u;
  const HcMaskCtx mc{lds + kPQ + 4 * v, lds + kPQ + 64 * MT};
  unsigned anym = 0u;
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
  // genotype block: a pad variant (column >= M) gets an offset beyond num_records and reads zeros
  const unsigned gbytes = (unsigned)((unsigned long long)M * (unsigned long long)ld * 8ull);  // < 2^31 (host checks)
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(uniform(gd.G), 0, gbytes, 0x00020000);
  const unsigned xbytes = (unsigned)((unsigned long long)nt.cols * (unsigned long long)ld * 8ull);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(uniform(nt.base), 0, xbytes, 0x00020000);
  const unsigned lane_off = (unsigned)(q * 32);  // 4 samples of 8 B per 16-lane row
  const unsigned col_bytes = (unsigned)((unsigned long long)ld * 8ull);
  unsigned vbase[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    const int col = c * 16 + v;
    vbase[c] = (col < M) ? (unsigned)col * col_bytes + lane_off : 0x80000000u;
  }
  const int xcol = (v <= d) ? v : d + 1;  // X_k, rr, or the zero column
  const unsigned xbase = (unsigned)xcol * col_bytes + lane_off;
  unsigned fx[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c) fx[c] = ((gd.pflip[c] >> v) & 1) ? 0x02020202u : 0u;

  d4_t accT[MT];
  i4_t accS[MT * (MT + 1) / 2];
  unsigned cs[MT], pk[MT][4];
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    accT[c] = d4_t{0.0, 0.0, 0.0, 0.0};
    cs[c] = 0;
  }
#pragma unroll
  for (int t = 0; t < MT * (MT + 1) / 2; ++t) accS[t] = i4_t{0, 0, 0, 0};
  HcBurden bu{0.0, 0.0, 0u, 0u};

  long long s = s_begin;
  const long long full = N >> 4;  // steps whose 16 samples all exist
  const long long s_fast_end = (s_end < full) ? s_end : full;
  constexpr int U = (DEPTH == 3) ? 12 : 4;
  const long long n_fast = (s_fast_end > s_begin) ? (s_fast_end - s_begin) / U : 0;
  if constexpr (DEPTH == 1) {
   if (n_fast > 0) {
    // Rolling refill (wide classes): ONE step buffer; as soon as a tile row of step s has been consumed its registers
    // are the destination of the same row of step s + 1, so every row's load has the other MT - 1 rows' work to land
    // in.  Half the ring registers of the two-buffer scheme (what lets MT = 5, 6 run two waves per SIMD).  The
    // null-model tile is double-buffered.
    unsigned voff[MT];
#pragma unroll
    for (int c = 0; c < MT; ++c) voff[c] = vbase[c] + (unsigned)(s_begin * 128);
    unsigned xoff = xbase + (unsigned)(s_begin * 128);
    u4_t glo[MT], ghi[MT], xlo[2], xhi[2];
#pragma unroll
    for (int c = 0; c < MT; ++c) {
      glo[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c], 0, NT ? 2 : 0));
      ghi[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c] + 16, 0, NT ? 2 : 0));
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
          unsigned p, m;
          hc_row<false>(glo[c], ghi[c], xv, accT[c], p, m, cs[c], fx[c], h, true);
          if (m) hc_note_masked(glo[c], ghi[c], m, mc.oa + 64 * c);
          if (hc_code3(p)) __hip_atomic_fetch_or(mc.flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          pk[c][u] = p | (m << 2);
          anym |= m;
          glo[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c] + (u + 1) * 128, 0, NT ? 2 : 0));
          ghi[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, voff[c] + (u + 1) * 128 + 16, 0, NT ? 2 : 0));
          __builtin_amdgcn_sched_barrier(0);
        }
        hc_finish<false>(h, xv, bu, 0xffffffffu);
        if (u == 3) hc_group_end<MT>(pk, accS, anym, lds, lane);
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
    for (int u = 0; u < DEPTH - 1; ++u) hc_issue<MT, NT>(f[u], rg, voff, rx, xoff, u * 128);
    for (long long it = 0; it < n_fast; ++it) {
      // step u lives in buffer u % DEPTH; the loads of step u + DEPTH - 1 are issued before step u is processed (the
      // last DEPTH - 1 issues belong to the next iteration; after the last iteration they are never used)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // (the scheduler must not hoist later steps' loads above this step's work: that is what the ring is for)
        hc_issue<MT, NT>(f[(u + DEPTH - 1) % DEPTH], rg, voff, rx, xoff, (u + DEPTH - 1) * 128);
        __builtin_amdgcn_sched_barrier(0);
        hc_step<MT, false>(f[u % DEPTH], u & 3, accT, pk, cs, fx, bu, true, 0xffffffffu, anym, mc);
        if ((u & 3) == 3) hc_group_end<MT>(pk, accS, anym, lds, lane);
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
    for (int c = 0; c < MT; ++c) pk[c][0] = pk[c][1] = pk[c][2] = pk[c][3] = 0u;
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
      hc_step<MT, true>(f, T, accT, pk, cs, fx, bu, valid, vmask, anym, mc);
    };
    one(0, s);
    one(1, s + 1);
    one(2, s + 2);
    one(3, s + 3);
    hc_group_end<MT>(pk, accS, anym, lds, lane);
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
          for (int i = 0; i < 4; ++i) out[(long long)(r * 16 + q * 4 + i) * Cp + col] = (double)accS[t][i];  // i32 map
        }
      }
  }
#pragma unroll
  for (int r = 0; r < MT; ++r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r * 16 + q + 4 * i;  // f64 C/D map
      if (M + v < Cp) out[(long long)row * Cp + M + v] = accT[r][i];
      if (M + 16 + v < Cp) out[(long long)row * Cp + M + 16 + v] = 0.0;
    }
  }
  // ---- column statistics: sum of H, min / max over the hard calls, masked count, OR / AND of the masked patterns --------
  long long cnt_w = ((s_end * 16 < N) ? s_end * 16 : N) - s_begin * 16;
  if (cnt_w < 0) cnt_w = 0;
  double* cst = gd.colstat + (long long)wpart * kHcColstatRows * gd.Mp;
  const unsigned wflag = lds[kPQ + 64 * MT];
  {
    int t = 0;
#pragma unroll
    for (int c = 0; c < MT; ++c) {
      unsigned sc = cs[c];
      sc += __shfl_xor(sc, 16, 64);
      sc += __shfl_xor(sc, 32, 64);
      // C_vv sits in lane v + 16 (v >> 2), accumulator element v & 3 of the diagonal tile; Q_vv likewise in the LDS
      // image: word (tile, v & 2 ? 1 : 0, that lane), half v & 1
      const i4_t dg = accS[t];
      const int sel = (lane & 3) == 0 ? dg[0] : ((lane & 3) == 1 ? dg[1] : ((lane & 3) == 2 ? dg[2] : dg[3]));
      const int diag = __shfl(sel, v + 16 * (v >> 2), 64);
      const int tq = MT * MT + t;
      const unsigned qw = lds[(tq * 2 + ((v >> 1) & 1)) * 64 + v + 16 * (v >> 2)];
      const long long nm = (long long)((qw >> (16 * (v & 1))) & 0xffffu);  // masked entries of column c * 16 + v
      t += MT - c;
      // C_vv = sum (H + 4m)^2 = sum H^2 + 16 nm
      const long long sm = (long long)sc, hh = (long long)diag - 16 * nm, n2 = (hh - sm) / 2, n1 = 2 * sm - hh,
                      n0 = cnt_w - n1 - n2 - nm;
      const double mn = n0 > 0 ? 0.0 : (n1 > 0 ? 1.0 : (n2 > 0 ? 2.0 : INFINITY));
      const double mx = n2 > 0 ? 2.0 : (n1 > 0 ? 1.0 : (n0 > 0 ? 0.0 : -INFINITY));
      if (lane < 16) {
        const int j = c * 16 + lane;
        cst[j] = (double)sm;
        cst[gd.Mp + j] = mn;
        cst[2 * gd.Mp + j] = mx;
        cst[3 * gd.Mp + j] = (double)nm;
        const unsigned* w = lds + kPQ + 64 * c + 4 * lane;
        unsigned long long* bits = reinterpret_cast<unsigned long long*>(cst);
        bits[4 * gd.Mp + j] = ((unsigned long long)w[1] << 32) | w[0];
        bits[5 * gd.Mp + j] = ((unsigned long long)w[3] << 32) | w[2];
      }
    }
  }
  if (gd.wflags && lane == 0) gd.wflags[wpart] = wflag;
  if ((wflag & 1u) && gd.pq) {  // the masked tiles of this wave-part (read by gene_assemble only when the flag says so)
    unsigned* dst = gd.pq + (long long)wpart * kPQ;
#pragma unroll 4
    for (int w = lane; w < kPQ; w += 64) dst[w] = lds[w];
  }
  // ---- burden partial sums: [test][U, c'c, count, c'X_0 .. c'X_{d-1}], test 0 = CMC, 1 = Zeggini ---------------------
  if (gd.bparts) {
    double ac = bu.a_cmc, az = bu.a_zeg;
    ac += __shfl_xor(ac, 16, 64);
    az += __shfl_xor(az, 16, 64);
    ac += __shfl_xor(ac, 32, 64);
    az += __shfl_xor(az, 32, 64);
    unsigned zz = bu.zz, cn = bu.cnt;
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

// One kernel per tile class and register budget: the engine picks per class (see suffstat_hc_config).
template <int MT, int DEPTH, int WAVES, bool NT>
__global__ __launch_bounds__(64, WAVES) void gene_suffstat_hc(const GeneDesc* __restrict__ genes, NullTile nt,
                                                              long long N, long long ld, int d) {
  __shared__ unsigned lds[hc_lds_words(MT)];
  const GeneDesc gd = genes[blockIdx.y];
  if (gd.MT != MT) return;
  suffstat_hc_body<MT, DEPTH, NT>(gd, nt, N, ld, d, lds);
}

// ---- block classification: is every entry of an N x M block exactly 0.0, 1.0 or 2.0? -------------------------------
// One pass at streaming rate (run once when a block is uploaded / registered, not per analysis step).  flag[0] is
// cleared when a value outside {0, 1, 2} is met (the caller sets it to 1 first).
template <int UNUSED = 0>  // (a template so that several translation units may include this header)
__global__ __launch_bounds__(256) void block_classify_kernel(const double* __restrict__ G, long long N, long long ld,
                                                             int M, int* __restrict__ flag) {
  const long long per_col = (N + 1) / 2;  // pairs of doubles
  const long long total = per_col * M;
  bool bad = false;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long j = idx / per_col, i = (idx % per_col) * 2;
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(G + j * ld + i);
    const unsigned long long a = p[0], b = (i + 1 < N) ? p[1] : 0ull;
    auto ok = [](unsigned long long x) {
      return x == 0ull || x == 0x3FF0000000000000ull || x == 0x4000000000000000ull;
    };
    bad |= !ok(a) || !ok(b);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicAnd(flag, 0);
}

}  // namespace rvt
