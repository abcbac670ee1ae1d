// rvtests_amd — sufficient statistics of HARD-CALL genotype blocks against a WEIGHTED (binary-trait) null model, computed by
// a WORKGROUP of four waves in INTEGER arithmetic throughout (round 4).  Same inputs and outputs as gene_suffstat_hcw
// (suffstat_hcw.hip.h; replaces SkatO.cpp:150-160 with V = diag(p(1-p)), LogisticRegressionScoreTest.cpp:260-263, the
// collapsers Model.cpp:73-89,115-130 and DataConsolidator.cpp:46-69,94-116), with the work divided differently:
//
//   * the one-wave kernel keeps ALL T = MT (MT + 1) / 2 Gram tiles x 3 pair accumulators of a gene AND the fp64 tiles of
//     G'V[X | rr] in one wave: 416-440 registers for M > 48 — one wave per SIMD, 40 KB of loads in flight per CU
//     (latency-bound at 4.9-5.4 TB/s), and a wave that needs a SIMD of its own: every 256-register wave of the per-gene
//     stages (p-values, assembly) that sits on a SIMD keeps such a wave out (0.46 of HBM live, 0.61-0.67 alone);
//   * here a workgroup is EIGHT waves in two roles.  Each of the four LOADER waves streams a different 64-sample slice of
//     ALL columns per iteration — the value tests, the packing to 2-bit integers, the byte sums and the per-sample burden
//     counts are per sample and stay with the wave that loaded it — and writes the packed operands (1 byte per genotype)
//     to LDS.  The four TILE waves multiply: each owns a share of the output tiles over the four slices — waves 4-6 the
//     Gram tiles G'VG (assigned by A-operand row: the row's byte-select operands d_p (x) g are built once per slice and
//     reused for the row's tiles), wave 7 the tiles G'V[X | res | v] and the burden sums: at most 6 tiles x 3 pairs = 72
//     accumulator registers per wave instead of 180 + 40.  ONE barrier per iteration, two operand buffers: the loaders
//     fill buffer i + 1 while the tile waves multiply buffer i, so a wave-part costs max(stream, multiply) instead of
//     their sum, the loaders never stop issuing loads (a ring of 2-4 steps per wave in flight), and the two roles share
//     every SIMD (waves w and w + 4 of a workgroup sit on SIMD w mod 4);
//   * G'V[X | res] and the burden sums c'V[X | res | v] go to the int8 matrix cores as well: the null-model tile
//     [vX_0 .. vX_{d-1} | res | v] is quantised ONCE per null model to six balanced base-128 digit planes per column with a
//     power-of-two scale per column (42 bits below twice the column's largest entry; rvt_set_null checks that no column's
//     largest entry exceeds 256 x its root mean square, else the model stays on gene_suffstat_hcw) and stored in operand
//     order.  One v_mfma_i32_16x16x64_i8 per plane, row tile and 64 samples replaces SIXTEEN fp64 instructions of 64 cycles
//     — the fp64 matrix pipe leaves the kernel — and every statistic is an exact integer of the quantised inputs:
//     bit-reproducible whatever the order of the sums;
//   * every class fits 256 registers: two waves per SIMD, 80-130 KB of loads in flight per CU, and any per-gene stage
//     can share the SIMD;
//   * MEAN-IMPUTED COLUMNS STAY ON THIS KERNEL (the one-wave kernel hands such a gene to the fp64 kernel: 0.33 of HBM).
//     G_j = H_j + mu_j m_j (integer H, 0/1 mask m, one mu per column: imputeGenotypeToMean, DataConsolidator.cpp:217-245):
//       G'VG = H'VH + P diag(mu) + diag(mu) P' + diag(mu) Q diag(mu),   P = H'Vm,  Q = m'Vm,
//       G'V[X | res] = H'V[X | res] + diag(mu) R,   R = m'V[X | res].
//     The dense tiles hold the H terms (a masked entry packs as H = 0).  P, Q and R are SPARSE — sums over the masked
//     entries — and are accumulated exactly, as 64-bit integers, by the wave that loaded the slice: for every masked
//     entry (sample i, column j) it walks sample i's row of packed integers and mask bits IN LDS (80 bytes, already there
//     for the Gram tiles) and adds V_i H_ik to P_jk, V_i to Q_jk and the fixed-point row i of the null tile to R_j
//     with integer atomics on the gene's own table in global memory (order-independent, hence bit-reproducible; measured
//     23 G atomics/s device-wide, the path needs ~1.5 G/s at 0.1 % missing calls).  No extra pass over G, no extra HBM
//     traffic.  That the masked entries of a column are bit-identical is verified with LDS atomics (OR / AND of the bit
//     patterns) exactly as in suffstat_hc.hip.h; a column where they are not sends the gene to the fp64 kernel.
//
// Wave-parts: one workgroup owns steps_per_wpart 16-sample steps (a multiple of 16); an iteration is 16 steps (4 loader waves
// x 4 steps = one int8 operand per wave and column tile).  The host cuts a gene into a quarter of the parts the one-wave
// kernels use (the partial-statistics image written per part and read by gene_assemble shrinks accordingly).
#pragma once
#include "suffstat_hcw.hip.h"

namespace rvt {

constexpr int kHcxNW = 4;          // loader waves = slices per iteration = tile waves (a workgroup is 2 kHcxNW waves)
constexpr int kHcxIterSteps = 16;  // steps per workgroup iteration (kHcxNW x 4)
constexpr int kHcxMaxMT = 5;
constexpr int kHcxNullCols = 16;   // the null tile is ONE 16-column operand: vX_0 .. vX_{d-1}, res, v, zeros
constexpr int kHcxStageCols = 8;   // non-zero null columns the kernel stages in LDS (d <= 6; wider models: gene_suffstat_hcw)

// Null-model operands of the kernel (built by rvt_set_null):
//   dq    digit planes of v in the order the tile waves read them: [group of 64 samples][q 4][pair 3][step 4] x 16 bytes =
//         (d_2j, d_2j+1, 2 d_2j, 2 d_2j+1) of the four samples 64 g + 16 T + 4 q + 0..3 (one dword each, a byte per sample):
//         768 bytes per slice, copied to LDS as they are
//   xq    digit planes of the null tile's ncols non-zero columns: [group of 64 samples][plane 0..5][q 0..3][column k < ncols]
//         x 16 bytes, byte 4 T + l = digit of sample 64 g + 16 T + 4 q + l — the operand of plane p is lane (k, q) <- entry
//         (p, q, k), zero for k >= ncols.  Both images are padded by four groups (an iteration is fetched as one range)
//   scale value of column k = integer x scale[k] (a power of two)
struct NullTileX {
  const unsigned char* dq;
  const unsigned char* xq;
  double scale[kHcxNullCols];
  int ncols;  // d + 2 non-zero columns (vX_0 .. vX_{d-1}, res, v); <= kHcxStageCols
};

// ---- tile assignment of the tile waves (index w = wave - 4).  0-2: Gram tiles by A-operand row; 3: the tiles G'V[X | res | v]
// of every row + the burden tile
struct HcxAsg {
  int nrows;
  int arow[2];
  int ncols[2];
  int col[2][5];
};
constexpr HcxAsg hcx_asg(int MT, int w) {
  switch (MT) {
    case 1:
      return w == 0 ? HcxAsg{1, {0, 0}, {1, 0}, {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}} : HcxAsg{0, {0, 0}, {0, 0}, {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}};
    case 2:
      return w == 0   ? HcxAsg{1, {0, 0}, {2, 0}, {{0, 1, 0, 0, 0}, {0, 0, 0, 0, 0}}}
             : w == 1 ? HcxAsg{1, {1, 0}, {1, 0}, {{1, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}}
                      : HcxAsg{0, {0, 0}, {0, 0}, {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}};
    case 3:
      return w == 0   ? HcxAsg{1, {0, 0}, {3, 0}, {{0, 1, 2, 0, 0}, {0, 0, 0, 0, 0}}}
             : w == 1 ? HcxAsg{1, {1, 0}, {2, 0}, {{1, 2, 0, 0, 0}, {0, 0, 0, 0, 0}}}
             : w == 2 ? HcxAsg{1, {2, 0}, {1, 0}, {{2, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}}
                      : HcxAsg{0, {0, 0}, {0, 0}, {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}};
    case 4:
      return w == 0   ? HcxAsg{1, {0, 0}, {4, 0}, {{0, 1, 2, 3, 0}, {0, 0, 0, 0, 0}}}
             : w == 1 ? HcxAsg{1, {1, 0}, {3, 0}, {{1, 2, 3, 0, 0}, {0, 0, 0, 0, 0}}}
             : w == 2 ? HcxAsg{2, {2, 3}, {2, 1}, {{2, 3, 0, 0, 0}, {3, 0, 0, 0, 0}}}
                      : HcxAsg{0, {0, 0}, {0, 0}, {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}};
    default:
      return w == 0   ? HcxAsg{1, {0, 0}, {5, 0}, {{0, 1, 2, 3, 4}, {0, 0, 0, 0, 0}}}
             : w == 1 ? HcxAsg{2, {1, 4}, {4, 1}, {{1, 2, 3, 4, 0}, {4, 0, 0, 0, 0}}}
             : w == 2 ? HcxAsg{2, {2, 3}, {3, 2}, {{2, 3, 4, 0, 0}, {3, 4, 0, 0, 0}}}
                      : HcxAsg{0, {0, 0}, {0, 0}, {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}};
  }
}
constexpr int hcx_ntiles(int MT, int w) {
  if (w == 3) return MT;  // (+ the burden tile, six unpaired plane sums)
  const HcxAsg a = hcx_asg(MT, w);
  return (a.nrows > 0 ? a.ncols[0] : 0) + (a.nrows > 1 ? a.ncols[1] : 0);
}
constexpr int hcx_max_tiles(int MT) {
  int m = 1;
  for (int w = 0; w < kHcxNW; ++w) m = hcx_ntiles(MT, w) > m ? hcx_ntiles(MT, w) : m;
  return m;
}

// ---- LDS of one workgroup (bytes) ------------------------------------------------------------------------------------
//   2 buffers x 4 slices x (MT + 1) x 64 lanes x 16 B: the int8 operand of every column tile (4 steps x 4 samples per lane),
//        then the burden operand (rows 0-3 = lanes v < 4: c_cmc, c_zeg, c_zeg^2 low 7 bits, c_zeg^2 >> 7 of the lane row's
//        samples; the other lanes hold zero)
//   per slice: 2 buffers x mk: MT x 64 lanes x 4 B: bit 4 T + l = entry (step T, sample l) of the lane's column is masked
//        (zero except between a masked entry and the end of its slice's treatment); the masked-entry list of
//        hcx_masked_slice; in the tail: one word per slice and buffer "the slice holds masked entries"
//   OR / AND words: MT x 64 x 4 B;  per column: masked-entry count, sum g, number of non-zero g: 3 x MT x 16 x 4 B;  4 words: flag (bit 0:
//        a masked entry was met, bit 1: an entry with code 3 = -inf), number of samples with a non-zero collapsed genotype
//   digit stage: 2 buffers x 4 slices x 768 B;  xq stage: 2 buffers x 4 slices x 6 planes x 4 lane rows x ncols x 16 B — the
//        weights' and the null tile's operands of an iteration (NullTileX::dq / xq as they lie in memory), fetched by LDS-DMA:
//        every loader wave its own slice's, at the start of its load phase (a tile wave that issues them is stalled for
//        thousands of cycles behind the loads the loaders keep the memory pipeline full with)
constexpr int hcx_slice_bytes(int MT) { return (MT + 1) * 1024; }
constexpr int hcx_buf_bytes(int MT) { return 2 * kHcxNW * hcx_slice_bytes(MT); }
constexpr int hcx_mk_bytes(int MT) { return MT * 256; }  // the mask words of one slice
constexpr int kHcxListCap = 48;   // entries per round: 12 bytes each (meta, V lo, V hi)
constexpr int kHcxUpdCap = 128;   // table updates collected per round: 12 bytes each (table index, value lo, value hi)
constexpr int hcx_list_bytes() { return 16 + kHcxListCap * 12 + kHcxUpdCap * 12; }
// per slice w: mask words of the two operand buffers, then the entry list of the tile wave that treats the slice
constexpr int hcx_wave_bytes(int MT) { return 2 * hcx_mk_bytes(MT) + hcx_list_bytes(); }
constexpr int hcx_tail_bytes(int MT) { return MT * 256 + 3 * MT * 64 + 16 + 2 * kHcxNW * 4; }
constexpr int kHcxSliceDg = 768;
constexpr int kHcxDgStage = 2 * kHcxNW * kHcxSliceDg;
constexpr int kHcxStageBuf = kHcxNW * kHcwPlanes * 4 * kHcxStageCols * 16;  // the null tile's operands of one iteration (at most)
constexpr int kHcxStageBytes = 2 * kHcxStageBuf;
constexpr int hcx_off_wave(int MT) { return hcx_buf_bytes(MT); }
constexpr int hcx_off_tail(int MT) { return hcx_off_wave(MT) + kHcxNW * hcx_wave_bytes(MT); }
constexpr int hcx_off_dg(int MT) { return hcx_off_tail(MT) + hcx_tail_bytes(MT); }
constexpr int hcx_off_stage(int MT) { return hcx_off_dg(MT) + kHcxDgStage; }
constexpr int hcx_lds_bytes(int MT) { return hcx_off_stage(MT) + kHcxStageBytes; }

// the masked-entry tables of a gene, 64-bit integers: P (Mp x Mp, row = the masked column; units 2^-42), Q (Mp x Mp, upper
// triangle; units 2^-42), R (Mp x 16: sum of the null tile's fixed-point rows over the column's masked samples; column
// k in units of NullTileX::scale[k])
constexpr size_t hcx_pq_entries(int Mp) { return 2 * (size_t)Mp * Mp + (size_t)Mp * kHcxNullCols; }

// one tile row of one step: hard-call test, packing, byte sums, burden hits.  A double is a hard call iff its low dword is
// zero and its high dword is 0, 0x3FF00000 or 0x40000000: hi + 0x00100000 then has no bit outside {20, 30} (see hc_row,
// suffstat_hc.hip.h).  The common path only asks whether ALL FOUR doubles of the lane are hard calls (`bad` = 0); which of
// them are not, the mask bytes and the clean-up of their codes are the rare path's (hcx_row_masked).  top: OR of the top
// bytes of every double the lane has met — bit 7 set in a byte says negative / -inf / NaN: the gene goes to the fp64 kernel.
// cs: running sum of g, cb: running count of non-zero g (their codes 1 = 01b and 2 = 10b hold one bit each): n2 = cs - cb,
// n1 = 2 cb - cs.
__device__ __forceinline__ void hcx_row(const u4_t& glo, const u4_t& ghi, unsigned& p, unsigned& bad, unsigned& top) {
  const unsigned w01 = __builtin_amdgcn_perm(glo[3], glo[1], 0x0c0c0703u);
  const unsigned w23 = __builtin_amdgcn_perm(ghi[3], ghi[1], 0x07030c0cu);
  const unsigned w = w01 | w23;
  top |= w;
  p = (w >> 5) & 0x03030303u;
  constexpr unsigned kAdd = 0x00100000u, kBits = 0xBFEFFFFFu;
  const unsigned i0 = ((glo[1] + kAdd) & kBits) | glo[0], i1 = ((glo[3] + kAdd) & kBits) | glo[2],
                 i2 = ((ghi[1] + kAdd) & kBits) | ghi[0], i3 = ((ghi[3] + kAdd) & kBits) | ghi[2];
  bad = (i0 | i1) | (i2 | i3);
}
// the lane holds an entry that is not a hard call: its mask bytes (0x01 per such entry); their codes are cleared
__device__ __forceinline__ unsigned hcx_row_masked(const u4_t& glo, const u4_t& ghi, unsigned& p) {
  constexpr unsigned kAdd = 0x00100000u, kBits = 0xBFEFFFFFu;
  const unsigned i0 = ((glo[1] + kAdd) & kBits) | glo[0], i1 = ((glo[3] + kAdd) & kBits) | glo[2],
                 i2 = ((ghi[1] + kAdd) & kBits) | ghi[0], i3 = ((ghi[3] + kAdd) & kBits) | ghi[2];
  auto one = [](unsigned x) { return x < 1u ? x : 1u; };  // v_min_u32
  const unsigned m = one(i0) | (one(i1) << 8) | (one(i2) << 16) | (one(i3) << 24);
  p &= ~(m * 3u);
  return m;
}
// sums and burden hits of the (cleaned) codes.  fx: 0x02 in every byte when the column is predicted flipped — (int)(2 - g) > 0
// <=> g != 2; a masked entry (code 0 after the clean-up) counts as 0 there, i.e. it would count for a flipped column: the
// rare path takes its bits out of h again (hcx_note)
__device__ __forceinline__ void hcx_sums(unsigned p, unsigned fx, unsigned& cs, unsigned& cb, unsigned& h) {
  cs = __builtin_amdgcn_sad_u8(p, 0u, cs);
  cb = (unsigned)__builtin_popcount(p) + cb;
  const unsigned t = p ^ fx;
  h += (t | (t >> 1)) & 0x01010101u;
}

// the rare path of one row-step — a lane that holds a masked entry: OR / AND of the bit patterns and the count of its column,
// the entry's bit in the slice's mask word of this lane (all in LDS: nothing of it lives in registers)
__device__ __forceinline__ void hcx_note(const u4_t& glo, const u4_t& ghi, unsigned m, int T, unsigned* mkw, unsigned* cmw,
                                         unsigned* oa, unsigned& anym, unsigned& h, unsigned fx) {
  anym = 1u;
  h -= m & (fx >> 1);  // (hcx_sums counts a cleared code in a flipped column: a masked entry never counts in the in-pass collapse)
  hc_note_masked(glo, ghi, m, oa);
  const unsigned nib = (m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xfu;
  __hip_atomic_fetch_or(mkw, nib << (4 * T), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_add(cmw, (unsigned)__builtin_popcount(nib), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The burden operand of one step from the per-sample variant counts (byte l of h = count of sample l of the lane row, the
// same in its 16 lanes): lane v = 0: c_cmc = (n > 0), v = 1: c_zeg = n, v = 2 / 3: the low 7 bits / the rest of n^2 (n <= 80).
__device__ __forceinline__ unsigned hcx_burden_bytes(unsigned h, int v, unsigned& cnt) {
  const unsigned cc = ((h + 0x7f7f7f7fu) >> 7) & 0x01010101u;  // n > 0 (n < 128)
  cnt += (unsigned)__builtin_popcount(cc);
  // n^2 of the four counts as packed 16-bit products (n <= 80: n^2 < 2^13)
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  const us2 a = __builtin_bit_cast(us2, h & 0x00ff00ffu), b = __builtin_bit_cast(us2, (h >> 8) & 0x00ff00ffu);
  const unsigned s02 = __builtin_bit_cast(unsigned, a * a), s13 = __builtin_bit_cast(unsigned, b * b);
  const unsigned lo = (s02 & 0x007f007fu) | ((s13 & 0x007f007fu) << 8);
  const unsigned hi = ((s02 >> 7) & 0x00ff00ffu) | (((s13 >> 7) & 0x00ff00ffu) << 8);
  return v == 0 ? cc : (v == 1 ? h : (v == 2 ? lo : (v == 3 ? hi : 0u)));
}

// Masked entries of the slice this wave has just written to LDS: P_jk += V_i H_ik, Q_jk += V_i m_ik (k >= j), R_jk += X_ik for
// every masked entry (sample i, column j).  The lanes that hold masked entries push them (column, lane row, step, sample,
// the sample's weight V_i) onto a small list in the wave's own LDS scratch, then ALL lanes share the items (entry, null
// column k | column k of the sample's row): the digits of the null tile's row from global memory resp. one LDS read of the
// packed integer and of the mask word, and the integer atomics.  A list that is full is worked off and refilled (a slice
// with many missing calls takes several rounds).  Clears the slice's mask words behind itself.  Run by the TILE wave of
// the slice's number after the iteration's barrier: the tile waves have time to spare, the loaders do not.
// dgb: the digits of v of this slice (LDS digit stage); xq_slice: the null tile's digit planes of this slice (LDS stage).
// (Inlined, and the pointers carry their address spaces.  As an out-of-line function taking generic pointers every LDS access
//  became a FLAT instruction — which queues in the vector-memory pipeline the loaders keep full: 12 000 cycles for five
//  entries — and its return waited for the table atomics to complete, a round trip to memory under full load.)
typedef __attribute__((address_space(3))) char hcx_lchar;
typedef __attribute__((address_space(3))) unsigned hcx_luint;
typedef __attribute__((address_space(1))) unsigned long long hcx_gull;
// apply the collected table updates (see hcx_masked_slice)
__device__ __forceinline__ void hcx_apply_updates(hcx_luint* cntw, hcx_gull* pq, int lane, int& U) {
  hcx_luint* upd = cntw + 4 + kHcxListCap * 3;
  asm volatile("" ::: "memory");
  for (int u = lane; u < U; u += 64)
    __hip_atomic_fetch_add(pq + upd[u * 3], ((unsigned long long)upd[u * 3 + 2] << 32) | upd[u * 3 + 1], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  U = 0;
}

template <int MT>
__device__ __forceinline__ void hcx_masked_slice(const hcx_lchar* pkb, const hcx_lchar* dgb, hcx_luint* mkb, hcx_luint* cntw, int lane,
                                              hcx_gull* pq, int Mp, const hcx_lchar* xq_slice, int ncols, int& U) {
  hcx_luint* list = cntw + 4;
  hcx_luint* upd = list + kHcxListCap * 3;
  const int v = lane & 15, q = lane >> 4;
  unsigned word[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c) word[c] = mkb[c * 64 + lane];
  // rank of this lane among the lanes of `mask` (wave-uniform compaction without atomics: the whole wave runs this code)
  auto rank_of = [&](unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
  };
  for (;;) {
    // ---- a round of entries: every lane that still holds masked entries hands over one per turn, until the list is full
    int E = 0;
    for (;;) {
      int c0 = -1;
#pragma unroll
      for (int c = MT - 1; c >= 0; --c) c0 = word[c] ? c : c0;
      const unsigned long long have = __builtin_amdgcn_ballot_w64(c0 >= 0);
      if (have == 0ull || E + __builtin_popcountll(have) > kHcxListCap) break;
      if (c0 >= 0) {
        unsigned wsel = 0u;
#pragma unroll
        for (int c = 0; c < MT; ++c) wsel = (c == c0) ? word[c] : wsel;
        const int b = __builtin_ctz(wsel);
#pragma unroll
        for (int c = 0; c < MT; ++c) word[c] = (c == c0) ? (word[c] & (word[c] - 1)) : word[c];
        const int T0 = b >> 2, l0 = b & 3;
        // V = sum_p d_p 128^(5 - p): the sample's weight in units of 2^-42 (what the six digit planes encode)
        long long V = 0;
#pragma unroll
        for (int jp = 0; jp < kHcwPairs; ++jp) {
          const hcx_luint* w = reinterpret_cast<const hcx_luint*>(dgb + ((q * kHcwPairs + jp) * 4 + T0) * 16);
          const int de = (int)(signed char)((w[0] >> (8 * l0)) & 0xffu), dod = (int)(signed char)((w[1] >> (8 * l0)) & 0xffu);
          V = V * 16384 + (long long)(de * 128 + dod);
        }
        const int idx = E + rank_of(have);
        list[idx * 3 + 0] = (unsigned)(c0 * 16 + v) | ((unsigned)q << 8) | ((unsigned)b << 12);
        list[idx * 3 + 1] = (unsigned)(unsigned long long)V;
        list[idx * 3 + 2] = (unsigned)((unsigned long long)V >> 32);
      }
      E += __builtin_popcountll(have);
    }
    if (E == 0) break;  // (nothing left)
    asm volatile("" ::: "memory");
    // ---- The table updates (index, value) are COLLECTED in LDS and applied by all lanes at once: an atomic instruction with
    // one or two active lanes per item costs as much as a full one, and every vector-memory instruction of this wave queues
    // behind the loads the loaders keep the memory pipeline full with (~1000 cycles each).  The collection lives on from
    // slice to slice (U: its fill, kept by the caller) and is applied when it is nearly full, and at the end of the wave-part.
    auto flush = [&]() {
      asm volatile("" ::: "memory");
      for (int u = lane; u < U; u += 64)
        __hip_atomic_fetch_add(pq + upd[u * 3], ((unsigned long long)upd[u * 3 + 2] << 32) | upd[u * 3 + 1], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      U = 0;
    };
    auto add = [&](bool pred, size_t index, unsigned long long val) {  // (called by the whole wave)
      const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
      const int n = __builtin_popcountll(m);
      if (U + n > kHcxUpdCap) flush();
      if (pred) {
        const int u = U + rank_of(m);
        upd[u * 3 + 0] = (unsigned)index;
        upd[u * 3 + 1] = (unsigned)val;
        upd[u * 3 + 2] = (unsigned)(val >> 32);
      }
      U += n;
    };
    // the null tile's rows of the entries' samples
    for (int item0 = 0; item0 < E * ncols; item0 += 64) {
      const int item = item0 + lane;
      const bool on = item < E * ncols;
      const int e = on ? item / ncols : 0, kk = item - e * ncols;
      const unsigned meta = list[e * 3];
      const int j = (int)(meta & 0xffu), q0 = (int)((meta >> 8) & 3u), b = (int)(meta >> 12);
      const hcx_lchar* src = xq_slice + (q0 * ncols + (on ? kk : 0)) * 16 + b;  // (byte 4 T0 + l0 = b)
      long long X = 0;
#pragma unroll
      for (int p = 0; p < kHcwPlanes; ++p) X = X * 128 + (long long)(signed char)src[p * 4 * ncols * 16];
      add(on, 2 * (size_t)Mp * Mp + (size_t)j * kHcxNullCols + kk, (unsigned long long)X);
    }
    constexpr int NKC = MT * 16;
    for (int item0 = 0; item0 < E * NKC; item0 += 128) {  // two items per lane and pass: their LDS reads overlap
      int j[2], k[2], b[2];
      unsigned hw[2], mw[2];
      unsigned long long Vu[2];
      bool on[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int item = item0 + lane + 64 * x;
        on[x] = item < E * NKC;
        const int e = on[x] ? item / NKC : 0;
        k[x] = item - e * NKC;
        const unsigned meta = list[e * 3];
        Vu[x] = ((unsigned long long)list[e * 3 + 2] << 32) | list[e * 3 + 1];
        j[x] = (int)(meta & 0xffu);
        b[x] = (int)(meta >> 12);
        const int q0 = (int)((meta >> 8) & 3u);
        const int slot = on[x] ? (k[x] >> 4) * 64 + (k[x] & 15) + 16 * q0 : 0;
        hw[x] = *reinterpret_cast<const hcx_luint*>(pkb + slot * 16 + (b[x] >> 2) * 4);
        mw[x] = mkb[slot];
      }
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        // (column k of the sample is either a hard call — its code goes to P — or masked — Q —, never both)
        const unsigned hval = (hw[x] >> (8 * (b[x] & 3))) & 3u;
        const bool isq = ((mw[x] >> b[x]) & 1u) != 0u;
        const size_t cell = (size_t)j[x] * Mp + k[x];
        add(on[x] && (hval != 0u || (isq && k[x] >= j[x])), isq ? (size_t)Mp * Mp + cell : cell, isq ? Vu[x] : Vu[x] * hval);
      }
    }
    if (U > kHcxUpdCap - 48) flush();
  }
#pragma unroll
  for (int c = 0; c < MT; ++c) mkb[c * 64 + lane] = 0u;
}

// two digit planes of one tile and one 64-sample operand: acc = 128 (A0'B0) + A1'B1 + acc, exactly, in int32
__device__ __forceinline__ void hcx_pair_step(i4_t& acc, const i4_t& a0, const i4_t& b0, const i4_t& a1, const i4_t& b1) {
  const i4_t z = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, i4_t{0, 0, 0, 0}, 0, 0, 0);
  i4_t a = acc;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = (int)(((unsigned)z[i] << 7) + (unsigned)a[i]);
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, a, 0, 0, 0);
}
__device__ __forceinline__ i4_t hcx_op(const u4_t& x) { return i4_t{(int)x[0], (int)x[1], (int)x[2], (int)x[3]}; }

// Gram tiles of wave W (0-2) over the kHcxNW slices of one buffer
template <int MT, int W, int NT>
__device__ __forceinline__ void hcx_gram(i4_t (&acc)[kHcwPairs][NT], const char* buf, const char* dgbuf, int lane) {
  constexpr HcxAsg A = hcx_asg(MT, W);
  const int q = lane >> 4;
#pragma unroll 1
  for (int s = 0; s < kHcxNW; ++s) {
    const char* slice = buf + s * hcx_slice_bytes(MT);
    const u4_t* pk = reinterpret_cast<const u4_t*>(slice) + lane;
    const u4_t* dg = reinterpret_cast<const u4_t*>(dgbuf + s * kHcxSliceDg) + q * (kHcwPairs * 4);
    int t = 0;
#pragma unroll
    for (int ri = 0; ri < A.nrows; ++ri) {
      const u4_t pr = pk[A.arow[ri] * 64];
      const unsigned pkr[4] = {pr[0], pr[1], pr[2], pr[3]};
      const HcwRow<MT> row(pkr);
#pragma unroll
      for (int j = 0; j < kHcwPairs; ++j) {
        i4_t a0, a1;
        {
          const u4_t d0 = dg[j * 4 + 0], d1 = dg[j * 4 + 1], d2 = dg[j * 4 + 2], d3 = dg[j * 4 + 3];
          a0 = i4_t{(int)__builtin_amdgcn_perm(d0[2], d0[0], row.sel[0]), (int)__builtin_amdgcn_perm(d1[2], d1[0], row.sel[1]),
                    (int)__builtin_amdgcn_perm(d2[2], d2[0], row.sel[2]), (int)__builtin_amdgcn_perm(d3[2], d3[0], row.sel[3])};
          a1 = i4_t{(int)__builtin_amdgcn_perm(d0[3], d0[1], row.sel[0]), (int)__builtin_amdgcn_perm(d1[3], d1[1], row.sel[1]),
                    (int)__builtin_amdgcn_perm(d2[3], d2[1], row.sel[2]), (int)__builtin_amdgcn_perm(d3[3], d3[1], row.sel[3])};
        }
#pragma unroll
        for (int ci = 0; ci < A.ncols[ri]; ++ci) {
          const i4_t b = hcx_op(pk[A.col[ri][ci] * 64]);
          hcx_pair_step(acc[j][t + ci], a0, b, a1, b);
        }
      }
      t += A.ncols[ri];
    }
  }
}

// Tile wave 3: the tiles H_r'V[X | res | v] of every row tile r (A = the packed integers as they are, B = digit plane p of the
// null tile, from the LDS stage) and the burden tile (A = the burden operand).  The burden operand's bytes reach 127 (the
// low part of a squared count), so its planes are NOT paired: 127 x 64 digits x 64 samples < 2^19 per operand and plane keeps
// a plain int32 sum exact over the 768 operands of the longest wave-part; shifted by 7 bits it would not.
template <int MT, int NT>
__device__ __forceinline__ void hcx_null_tiles(i4_t (&acc)[kHcwPairs][NT], i4_t (&accb)[kHcwPlanes], const char* buf, int lane,
                                               const char* stage, int xslot, int plane_bytes) {
#pragma unroll 1
  for (int s = 0; s < kHcxNW; ++s) {
    const char* slice = buf + s * hcx_slice_bytes(MT);
    const u4_t* pk = reinterpret_cast<const u4_t*>(slice) + lane;
    const char* st = stage + s * (kHcwPlanes * plane_bytes) + (xslot < 0 ? 0 : xslot);  // (slice s: planes of 4 ncols x 16 B)
    const i4_t ab = hcx_op(pk[MT * 64]);
#pragma unroll
    for (int j = 0; j < kHcwPairs; ++j) {
      u4_t q0 = *reinterpret_cast<const u4_t*>(st + (2 * j) * plane_bytes);
      u4_t q1 = *reinterpret_cast<const u4_t*>(st + (2 * j + 1) * plane_bytes);
      if (xslot < 0) q0 = q1 = u4_t{0u, 0u, 0u, 0u};  // (a zero column of the null tile)
      const i4_t b0 = hcx_op(q0), b1 = hcx_op(q1);
#pragma unroll
      for (int r = 0; r < MT; ++r) {
        const i4_t a = hcx_op(pk[r * 64]);
        hcx_pair_step(acc[j][r], a, b0, a, b1);
      }
      accb[2 * j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ab, b0, accb[2 * j], 0, 0, 0);
      accb[2 * j + 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ab, b1, accb[2 * j + 1], 0, 0, 0);
    }
  }
}

// LDS-DMA of `bytes` contiguous bytes (a multiple of 16) from global memory to LDS, 64 lanes x 16 bytes per instruction, lane-linear
__device__ __forceinline__ void hcx_dma(char* dst, const unsigned char* src, int bytes, int lane) {
  for (int off = 0; off < bytes; off += 1024)
    if (off + lane * 16 < bytes)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + off), 16, 0, 0);
}

// pair tiles -> the integer they encode, as a double (one rounding when it exceeds 2^53): p0 2^28 + p1 2^14 + p2
__device__ __forceinline__ double hcx_pairs_value(int p0, int p1, int p2) {
  const long long x = ((long long)p0 << 28) + ((long long)p1 << 14) + (long long)p2;
  return (double)x;
}

// ring depth of a loader wave per tile class: steps in flight (1 .. 4; 3: the loop body is three iterations)
constexpr int hcx_ring(int MT) { return MT <= 3 ? 4 : 2; }

template <int MT>
__device__ __forceinline__ void suffstat_hcx_body(const GeneDesc& gd, const NullTileX& nt, long long N, long long ld, int d,
                                                  char* lds) {
  constexpr int NT = hcx_max_tiles(MT);
  constexpr int RING = hcx_ring(MT);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // 0-3: loader waves (slice w of every iteration), 4-7:
                                                                          // tile waves; wave-uniform, i.e. scalar branches
  const int v = lane & 15, q = lane >> 4;
  const int wpart = blockIdx.x;
  if (wpart >= gd.n_wparts) return;
  const long long nsteps = ld >> 4;
  const long long s_begin = (long long)wpart * gd.steps_per_wpart;
  long long s_end = s_begin + gd.steps_per_wpart;
  if (s_end > nsteps) s_end = nsteps;
  if (s_begin >= s_end) return;  // (uniform over the workgroup)
  const int M = gd.M;
  unsigned* const oa_all = reinterpret_cast<unsigned*>(lds + hcx_off_tail(MT));
  unsigned* const cm_all = oa_all + MT * 64;     // masked count, then sum g, then the number of non-zero g: [3][MT * 16]
  unsigned* const flagw = cm_all + 3 * MT * 16;  // [0] flags, [1] samples with a non-zero collapsed genotype
  unsigned* const mflag = flagw + 4;             // [buffer][slice]: the slice holds masked entries
  char* const stage = lds + hcx_off_stage(MT);
  char* const dgstage = lds + hcx_off_dg(MT);
  // OR words = 0, AND words = ~0, counts = 0, flags = 0; the loaders' mask words and the burden operand of every slice = 0
  for (int x = threadIdx.x; x < MT * 64; x += 2 * kHcxNW * 64) oa_all[x] = (x & 2) ? 0xffffffffu : 0u;
  for (int x = threadIdx.x; x < 3 * MT * 16 + 4 + 2 * kHcxNW; x += 2 * kHcxNW * 64) cm_all[x] = 0u;
  for (int x = threadIdx.x; x < kHcxNW * 2 * MT * 64; x += 2 * kHcxNW * 64)
    reinterpret_cast<unsigned*>(lds + hcx_off_wave(MT) + (x / (2 * MT * 64)) * hcx_wave_bytes(MT))[x % (2 * MT * 64)] = 0u;
  for (int x = threadIdx.x; x < 2 * kHcxNW * 256; x += 2 * kHcxNW * 64)
    reinterpret_cast<unsigned*>(lds + (x / 256) * hcx_slice_bytes(MT) + MT * 1024)[x % 256] = 0u;
  __syncthreads();

#ifdef HCX_PROF
  long long prof[4] = {0, 0, 0, 0};  // cycles: loaders: load phase, barrier wait, masked entries | tile waves: barrier wait, tiles
#define HCX_NOW() ((long long)__builtin_readcyclecounter())
#define HCX_TICK(k, t0) prof[k] += HCX_NOW() - (t0)
#define HCX_DUMP()                                                                                                       \
  if (gd.dbg_cmc && lane == 0)                                                                                           \
    for (int k = 0; k < 4; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(gd.dbg_cmc) + w * 4 + k, (unsigned long long)prof[k])
#else
#define HCX_NOW() 0ll
#define HCX_TICK(k, t0)
#define HCX_DUMP()
#endif
  const long long full = N >> 4;  // steps whose 16 samples all exist
  const long long s_fast_end = (s_end < full) ? s_end : full;
  const long long n_fast = (s_fast_end > s_begin) ? (s_fast_end - s_begin) / kHcxIterSteps : 0;
  const long long n_iter = (s_end - s_begin + kHcxIterSteps - 1) / kHcxIterSteps;
  double* out = gd.parts + (long long)wpart * gd.Mp * gd.Cp;
  const int Cp = gd.Cp;

  if (w >= kHcxNW) {
    // ================================================ tile waves ================================================
    const int tw = w - kHcxNW;
    i4_t acc[kHcwPairs][NT];
#pragma unroll
    for (int j = 0; j < kHcwPairs; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[j][t] = i4_t{0, 0, 0, 0};
    int n_upd = 0;          // masked-entry table updates collected in LDS and not yet applied (hcx_masked_slice)
    i4_t accb[kHcwPlanes];  // tile wave 3: the burden tile, one sum per digit plane
#pragma unroll
    for (int p = 0; p < kHcwPlanes; ++p) accb[p] = i4_t{0, 0, 0, 0};
    // tile wave 3's view of the null tile: operand lane (k, q) reads stage entry q ncols + k of a plane
    const int ncols = nt.ncols;
    const int plane_bytes = 4 * ncols * 16;
    const int xslot = (v < ncols) ? (q * ncols + v) * 16 : -1;
    for (long long it = 0; it < n_iter; ++it) {
      const char* buf = lds + (int)(it & 1) * (kHcxNW * hcx_slice_bytes(MT));
      const char* dgbuf = dgstage + (int)(it & 1) * (kHcxNW * kHcxSliceDg);
      const long long t_a = HCX_NOW();
      __syncthreads();  // the loaders' operands and the stages of this iteration have arrived
      HCX_TICK(0, t_a);
      const long long t_b = HCX_NOW();
      switch (tw) {
        case 0: hcx_gram<MT, 0, NT>(acc, buf, dgbuf, lane); break;
        case 1: hcx_gram<MT, 1, NT>(acc, buf, dgbuf, lane); break;
        case 2: hcx_gram<MT, 2, NT>(acc, buf, dgbuf, lane); break;
        default: hcx_null_tiles<MT, NT>(acc, accb, buf, lane, stage + (int)(it & 1) * kHcxStageBuf, xslot, plane_bytes); break;
      }
      HCX_TICK(1, t_b);
      // The masked entries of slice tw -> P / Q / R (rare).  Everything it reads is in LDS and stays until the loaders refill
      // the buffers two iterations later: the slice's operands and mask words, the digits of v, the null tile's planes.
      if (mflag[(it & 1) * kHcxNW + tw]) {
        const long long t_c = HCX_NOW();
        char* ws = lds + hcx_off_wave(MT) + tw * hcx_wave_bytes(MT);
        unsigned* mkb = reinterpret_cast<unsigned*>(ws + (int)(it & 1) * hcx_mk_bytes(MT));
        if (gd.pqw)
          hcx_masked_slice<MT>((const hcx_lchar*)(buf + tw * hcx_slice_bytes(MT)), (const hcx_lchar*)(dgbuf + tw * kHcxSliceDg),
                               (hcx_luint*)mkb, (hcx_luint*)(ws + 2 * hcx_mk_bytes(MT)), lane, (hcx_gull*)gd.pqw, gd.Mp,
                               (const hcx_lchar*)(stage + (int)(it & 1) * kHcxStageBuf + tw * (kHcwPlanes * plane_bytes)), ncols, n_upd);
        else
          for (int x = lane; x < MT * 64; x += 64) mkb[x] = 0u;
        if (lane == 0) {
          mflag[(it & 1) * kHcxNW + tw] = 0u;
          __hip_atomic_fetch_or(flagw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        HCX_TICK(2, t_c);
      }
    }
    if (n_upd > 0)
      hcx_apply_updates((hcx_luint*)(lds + hcx_off_wave(MT) + tw * hcx_wave_bytes(MT) + 2 * hcx_mk_bytes(MT)), (hcx_gull*)gd.pqw, lane, n_upd);
    HCX_DUMP();
    // ---- partial tiles: element (row, col) -> parts[row * Cp + col], the layout gene_assemble reduces -------------
    auto store_gram = [&](auto wtag) {
      constexpr int W = decltype(wtag)::value;
      constexpr HcxAsg A = hcx_asg(MT, W);
      int t = 0;
#pragma unroll
      for (int ri = 0; ri < A.nrows; ++ri)
#pragma unroll
        for (int ci = 0; ci < A.ncols[ri]; ++ci, ++t) {
          const int r = A.arow[ri], c = A.col[ri][ci];
          const int col = c * 16 + v;
          if (col < M) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              out[(long long)(r * 16 + q * 4 + i) * Cp + col] = hcx_pairs_value(acc[0][t][i], acc[1][t][i], acc[2][t][i]) * 0x1p-42;
          }
        }
    };
    if (tw == 0) store_gram(std::integral_constant<int, 0>{});
    if (tw == 1) store_gram(std::integral_constant<int, 1>{});
    if (tw == 2) store_gram(std::integral_constant<int, 2>{});
    __syncthreads();  // (the loaders' counts are in LDS)
    if (tw == 3) {
      // G'V[X | res] columns of the partial matrix (i32 C/D map: lane (v, q), element i = row 4 q + i, column v)
      const double sc = nt.scale[v];
#pragma unroll
      for (int r = 0; r < MT; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = r * 16 + q * 4 + i;
          const double x = hcx_pairs_value(acc[0][r][i], acc[1][r][i], acc[2][r][i]) * sc;
          if (M + v < Cp) out[(long long)row * Cp + M + v] = (v <= d) ? x : 0.0;  // (the v column is not part of R)
          if (M + 16 + v < Cp) out[(long long)row * Cp + M + 16 + v] = 0.0;
        }
      }
      // ---- burden partial sums: [test][U, c'Vc, count, c'VX_0 .. c'VX_{d-1}], test 0 = CMC, 1 = Zeggini.  The burden tile's
      // rows 0-3 (lanes q = 0, elements 0-3): c_cmc, c_zeg, low / high part of c_zeg^2 against the null column v
      if (gd.bparts && lane < 16) {
        auto planes = [&](int i) {  // sum_p plane_p 128^(5 - p): |.| < 2^19 768 2^35 (exact in 64 bits)
          long long x = 0;
#pragma unroll
          for (int p = 0; p < kHcwPlanes; ++p) x = x * 128 + (long long)accb[p][i];
          return x;
        };
        const double ac = (double)planes(0) * sc;
        const double az = (double)planes(1) * sc;
        const double azz = (double)(planes(2) + 128 * planes(3)) * sc;
        const double cn = (double)flagw[1];
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
          bp[2] = cn;
          bp[rl + 2] = cn;
        }
      }
    }
    if (tw == 0) {
      // ---- column statistics (six rows as suffstat_hc.hip.h writes them) --------------------------------------------
      long long cnt_w = ((s_end * 16 < N) ? s_end * 16 : N) - s_begin * 16;
      if (cnt_w < 0) cnt_w = 0;
      double* cst = gd.colstat + (long long)wpart * kHcColstatRows * gd.Mp;
      for (int j = lane; j < MT * 16; j += 64) {
        const int c = j >> 4, l = j & 15;
        const long long nm = cm_all[j], sm = cm_all[MT * 16 + j], nz = cm_all[2 * MT * 16 + j];  // masked, sum g, # g != 0
        const long long n2 = sm - nz, n1 = 2 * nz - sm, n0 = cnt_w - n1 - n2 - nm;
        const double mn = n0 > 0 ? 0.0 : (n1 > 0 ? 1.0 : (n2 > 0 ? 2.0 : INFINITY));
        const double mx = n2 > 0 ? 2.0 : (n1 > 0 ? 1.0 : (n0 > 0 ? 0.0 : -INFINITY));
        cst[j] = (double)sm;
        cst[gd.Mp + j] = mn;
        cst[2 * gd.Mp + j] = mx;
        cst[3 * gd.Mp + j] = (double)nm;
        const unsigned* ow = oa_all + 64 * c + 4 * l;
        unsigned long long* bits = reinterpret_cast<unsigned long long*>(cst);
        bits[4 * gd.Mp + j] = ((unsigned long long)ow[1] << 32) | ow[0];
        bits[5 * gd.Mp + j] = ((unsigned long long)ow[3] << 32) | ow[2];
      }
      if (gd.wflags && lane == 0) gd.wflags[wpart] = flagw[0] & 3u;
    }
    return;
  }

  // ================================================== loader waves ==================================================
  char* const wscratch = lds + hcx_off_wave(MT) + w * hcx_wave_bytes(MT);  // this slice's mask words (two buffers)
  unsigned* const oa = oa_all + 4 * v;  // this lane's column of row tile c: oa + 64 c
  unsigned* const cmw = cm_all + v;     //                                     cmw + 16 c
  unsigned* const mkw0 = reinterpret_cast<unsigned*>(wscratch) + lane;

  auto uniform = [](const void* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
  };
  const unsigned gbytes = (unsigned)((unsigned long long)M * (unsigned long long)ld * 8ull);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(uniform(gd.G), 0, gbytes, 0x00020000);
  const unsigned lane_off = (unsigned)(q * 32);
  const unsigned col_bytes = (unsigned)((unsigned long long)ld * 8ull);
  // Column tile c of the lane: byte offset (c 16 + v) col_bytes + lane_off.  The tiles 0 .. MT - 2 are full (every column
  // exists), so ONE per-lane offset serves them, with the tile's 16 col_bytes as the instruction's scalar offset; the last
  // tile has its own offset, beyond num_records for a pad column (reads zeros).
  const unsigned vfull = (unsigned)v * col_bytes + lane_off;
  const unsigned vlast = ((MT - 1) * 16 + v < M) ? (unsigned)((MT - 1) * 16 + v) * col_bytes + lane_off : 0x80000000u;
  const unsigned tile_bytes = (unsigned)__builtin_amdgcn_readfirstlane((int)(16u * col_bytes));
  auto gload = [&](int c, unsigned off_full, unsigned off_last, int imm) {
    return (c == MT - 1) ? __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, off_last + imm, 0, 0))
                         : __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, off_full + imm, c * tile_bytes, 0));
  };
  int fxb = 0;  // bit c: the lane's column of tile c is predicted flipped
#pragma unroll
  for (int c = 0; c < MT; ++c) fxb |= (int)((gd.pflip[c] >> v) & 1) << c;
  auto fxof = [&](int c) { return (unsigned)__builtin_amdgcn_sbfe(fxb, c, 1) & 0x02020202u; };

  unsigned cs[MT], cb[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c) cs[c] = cb[c] = 0u;
  unsigned top = 0u, anym = 0u, cnt = 0u;

  // The digits of v and the null tile's planes of this wave's slice of iteration i -> the stage buffers i & 1 (LDS-DMA: no
  // registers, three instructions).  Issued at the start of the load phase, i.e. before the ring's refills: see the wait
  // in front of the barrier.  The buffers were last read two iterations ago.
  const int xq_group = kHcwPlanes * 4 * nt.ncols * 16;
  auto fetch_operands = [&](long long i) {
    const long long g = ((s_begin + i * kHcxIterSteps) >> 2) + w;
    hcx_dma(dgstage + (int)(i & 1) * (kHcxNW * kHcxSliceDg) + w * kHcxSliceDg, nt.dq + g * kHcxSliceDg, kHcxSliceDg, lane);
    hcx_dma(stage + (int)(i & 1) * kHcxStageBuf + w * xq_group, nt.xq + g * xq_group, xq_group, lane);
  };
  // a slice with masked entries: tell the tile wave that treats it (before the iteration's barrier)
  auto flag_masked = [&](long long i) {
    if (__builtin_amdgcn_ballot_w64(anym != 0u) != 0ull) {
      if (lane == 0) mflag[(i & 1) * kHcxNW + w] = 1u;
      anym = 0u;
    }
  };
  long long it = 0;
  constexpr int UNR = (RING == 3) ? 3 : 1;  // iterations per unrolled body: the ring index of a step must be a constant
  const long long n_body = n_fast / UNR;
  if (n_body > 0) {
    // Rolling refill: a ring of RING genotype step buffers; as soon as a tile row of a step has been consumed its registers
    // are the destination of the same row of the wave's step RING steps ahead (across iteration boundaries: the wave's
    // steps 4 i + u lie 16 steps apart from iteration to iteration).  Iterations left over by the unrolled body (RING = 3:
    // up to two, at the end of a gene only — the host cuts wave-parts in multiples of 3 iterations) take the ragged loop.
    unsigned offF = vfull + (unsigned)((s_begin + 4 * w) * 128), offL = vlast + (unsigned)((s_begin + 4 * w) * 128);
    u4_t glo[RING][MT], ghi[RING][MT];
    auto gload2 = [&](int c, int n_ahead, int half) {  // the wave's step n_ahead (counted from the body's first step)
      const int imm = (n_ahead & 3) * 128 + half * 16;
      const unsigned so = (unsigned)((n_ahead >> 2) * kHcxIterSteps * 128);  // whole iterations: a scalar offset
      return (c == MT - 1) ? __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, offL + imm, so, 0))
                           : __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, offF + imm, c * tile_bytes + so, 0));
    };
#pragma unroll
    for (int r = 0; r < RING; ++r)
#pragma unroll
      for (int c = 0; c < MT; ++c) {
        glo[r][c] = gload2(c, r, 0);
        ghi[r][c] = gload2(c, r, 1);
      }
    for (long long body = 0; body < n_body; ++body) {
#pragma unroll
      for (int k = 0; k < UNR; ++k, ++it) {
        const long long t_a = HCX_NOW();
        char* slice = lds + (int)(it & 1) * (kHcxNW * hcx_slice_bytes(MT)) + w * hcx_slice_bytes(MT);
        unsigned* pkw = reinterpret_cast<unsigned*>(slice) + lane * 4;
        unsigned* mkw = mkw0 + (int)(it & 1) * (MT * 64);
        fetch_operands(it);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = 4 * k + u, rb = n % RING;  // this step's ring buffer
          unsigned h = 0;
#pragma unroll
          for (int c = 0; c < MT; ++c) {
            unsigned bad, p;
            const unsigned fx = fxof(c);
            hcx_row(glo[rb][c], ghi[rb][c], p, bad, top);
            if (bad) {
              const unsigned m = hcx_row_masked(glo[rb][c], ghi[rb][c], p);
              hcx_note(glo[rb][c], ghi[rb][c], m, u, mkw + 64 * c, cmw + 16 * c, oa + 64 * c, anym, h, fx);
            }
            hcx_sums(p, fx, cs[c], cb[c], h);
            pkw[c * 256 + u] = p;
            glo[rb][c] = gload2(c, n + RING, 0);
            ghi[rb][c] = gload2(c, n + RING, 1);
            __builtin_amdgcn_sched_barrier(0);
          }
          h = row16_sum(h);
          const unsigned bb = hcx_burden_bytes(h, v, cnt);
          if (v < 4) pkw[MT * 256 + u] = bb;
          __builtin_amdgcn_sched_barrier(0);
        }
        flag_masked(it);
        HCX_TICK(0, t_a);
        const long long t_b = HCX_NOW();
        // (loads return in order: with only the ring's RING x MT x 2 loads outstanding, the DMA issued before them has landed)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING * MT * 2) : "memory");
        __syncthreads();
        HCX_TICK(1, t_b);
      }
      offF += UNR * kHcxIterSteps * 128;
      offL += UNR * kHcxIterSteps * 128;
    }
  }
  for (; it < n_iter; ++it) {  // ragged end: every step loaded from a clamped position and masked
    const long long s0 = s_begin + it * kHcxIterSteps + 4 * w;
    char* slice = lds + (int)(it & 1) * (kHcxNW * hcx_slice_bytes(MT)) + w * hcx_slice_bytes(MT);
    unsigned* pkw = reinterpret_cast<unsigned*>(slice) + lane * 4;
    unsigned* mkw = mkw0 + (int)(it & 1) * (MT * 64);
    fetch_operands(it);
#pragma unroll 1
    for (int u = 0; u < 4; ++u) {
      const long long su = s0 + u;
      const bool valid = su < s_end;
      const long long sc = valid ? su : s_end - 1;
      const unsigned so = (unsigned)(sc * 128);
      unsigned vmask = 0u;
      const long long smp = sc * 16 + q * 4;
#pragma unroll
      for (int l = 0; l < 4; ++l) vmask |= (valid && smp + l < N) ? (0xffu << (8 * l)) : 0u;
      unsigned h = 0;
#pragma unroll
      for (int c = 0; c < MT; ++c) {
        const u4_t glo = gload(c, vfull + so, vlast + so, 0), ghi = gload(c, vfull + so, vlast + so, 16);
        unsigned bad, p;
        const unsigned fx = fxof(c);
        hcx_row(glo, ghi, p, bad, top);
        p = valid ? p : 0u;
        bad = valid ? bad : 0u;
        if (bad) {
          const unsigned m = hcx_row_masked(glo, ghi, p);
          hcx_note(glo, ghi, m, u, mkw + 64 * c, cmw + 16 * c, oa + 64 * c, anym, h, fx);
        }
        hcx_sums(p, valid ? fx : 0u, cs[c], cb[c], h);
        pkw[c * 256 + u] = p;
      }
      h = row16_sum(h) & vmask;
      const unsigned bb = hcx_burden_bytes(h, v, cnt);
      if (v < 4) pkw[MT * 256 + u] = bb;
    }
    flag_masked(it);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  HCX_DUMP();
  // ---- byte sums, counts and flags of the loader waves meet in LDS (integers: any order) ---------------------------------
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    unsigned sc = cs[c], sq = cb[c];
    sc += __shfl_xor(sc, 16, 64);
    sq += __shfl_xor(sq, 16, 64);
    sc += __shfl_xor(sc, 32, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (lane < 16) {
      __hip_atomic_fetch_add(cm_all + (1 * MT + c) * 16 + lane, sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(cm_all + (2 * MT + c) * 16 + lane, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  {
    // (every lane of a 16-lane row counted the row's samples: lane 0 of each row speaks for it)
    unsigned cn = (v == 0) ? cnt : 0u;
    cn += __shfl_xor(cn, 16, 64);
    cn += __shfl_xor(cn, 32, 64);
    const bool c3 = __builtin_amdgcn_ballot_w64((top & 0x80808080u) != 0u) != 0ull;  // a negative value, -inf or NaN was met
    if (lane == 0) {
      __hip_atomic_fetch_add(flagw + 1, cn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (c3) __hip_atomic_fetch_or(flagw, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();  // (pairs with the tile waves' barrier before they read the counts)
}

template <int MT>
__global__ __launch_bounds__(2 * kHcxNW * 64, 2) void gene_suffstat_hcx(const GeneDesc* __restrict__ genes, NullTileX nt,
                                                                         long long N, long long ld, int d) {
  const GeneDesc gd = genes[blockIdx.y];
  if (gd.MT != MT) return;
  __shared__ __attribute__((aligned(16))) char lds[hcx_lds_bytes(MT)];
  suffstat_hcx_body<MT>(gd, nt, N, ld, d, lds);
}

// Every tile class in ONE launch (the engine's default): the workgroup takes the body of its gene's class (a scalar branch —
// the descriptor comes through scalar loads).  A batch's genes are sorted widest class first, so the launch hands out its
// longest workgroups first and ends on the shortest ones: one tail per batch instead of one per class, no launch gaps
// between the classes.  Registers and LDS are those of the widest class (every class fills a CU with one workgroup anyway).
template <int TOP>  // (the widest class compiled in; a template so that the header can sit in several translation units)
__global__ __launch_bounds__(2 * kHcxNW * 64, 2) void gene_suffstat_hcx_any(const GeneDesc* __restrict__ genes, NullTileX nt,
                                                                             long long N, long long ld, int d) {
  const GeneDesc gd = genes[blockIdx.y];
  __shared__ __attribute__((aligned(16))) char lds[hcx_lds_bytes(TOP)];
  switch (gd.MT) {
    case 1: suffstat_hcx_body<1>(gd, nt, N, ld, d, lds); break;
    case 2: suffstat_hcx_body<2>(gd, nt, N, ld, d, lds); break;
    case 3: suffstat_hcx_body<3>(gd, nt, N, ld, d, lds); break;
    case 4: suffstat_hcx_body<4>(gd, nt, N, ld, d, lds); break;
    case 5: suffstat_hcx_body<5>(gd, nt, N, ld, d, lds); break;
    default: break;
  }
}

}  // namespace rvt
