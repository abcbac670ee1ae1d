// rvtests_amd — sufficient statistics of DOSAGE blocks whose entries are FLOAT-PRECISION values (BGEN genotype
// probabilities: the reference forms the dosage from float probabilities, src/BGenGenotypeExtractor.cpp:413-478; any
// imputation output read as float), against an unweighted (quantitative-trait) null model — on the int8 matrix cores, exactly.
//
// Why there is an exact integer form.  A probability of a B-bit BGEN block is float(v) * float(1 / (2^B - 1)): at least
// 2^-B, so its float ulp is at least 2^-(B + 23); the dosage p1 + 2 p2 (and 1 - p0 - p1 sums) is a multiple of that ulp.  For
// 8-bit blocks every dosage is an integer multiple of 2^-31, and a float-precision dosage of any origin is a multiple of
// 2^-37 as soon as it is 0 or at least 2^-14.  So K = g 2^37 is an INTEGER below 2^38 (g <= 2), G'G = K'K / 2^74, and K'K is
// computed exactly:
//   * K is read off the double without a conversion: t = g + 2^15 has ulp 2^-37, its mantissa IS K; (t - 2^15 == g) is the
//     test that g is on the grid (it fails for a dosage printed with decimals, for a 16-bit BGEN probability below 2^-14, for
//     NaN), the high dword <= 0x40000000 the range test (0 <= g <= 2 and a little: K + bias fits five bytes).  A block that
//     fails is handed back to the fp64 kernel by gene_flags_hc_kernel, as with the lattice kernel (suffstat_lat.hip.h);
//   * K = sum_p s_p 256^p with five BALANCED base-256 digits s_p in [-128, 127]: they are the bytes of K + 0x8080808080 with
//     their top bits flipped — ONE 64-bit integer addition produces all five (the carries of the addition are the carries of
//     the balanced representation), where base-128 digits take a shift and a mask each;
//   * a tile of K'K is sum_{p, q} 256^(p + q) S_p' S_q: twenty-five v_mfma_i32_16x16x64_i8 per tile pair and 64 samples,
//     accumulated per order p + q in nine int32 tiles (|s_p s_q| <= 2^14: a wave-part of up to 8 192 samples stays below
//     2^31 with five products per order) and combined in fp64 once per wave-part, from the top order down (one rounding
//     where the fp64 kernel rounds every product and every sum);
//   * G'[X | res] goes the same way against five balanced base-256 digit planes of the null-model columns (quantised once
//     per null model with a power-of-two scale per column, rvt_set_null), the burden sums against the same planes.
// Cost: 25 int8 instructions of 16 cycles per tile pair and 64 samples against 16 fp64 instructions of 64 cycles — the matrix
// pipe that bounds gene_suffstat_mfma at 0.42 of the HBM rate is 0.4 times as busy.
//
// Structure: the workgroup-cooperative form of suffstat_hcx.hip.h — four loader waves (each a 32-sample slice of ALL columns
// per iteration: grid test, digits, byte sums, min / max, the per-sample burden counts) and four tile waves (nine order
// accumulators per tile pair are 36 registers: five pairs per wave for M in 65..80; the tile pairs
// (r, c), c = MT standing for the null tile, dealt out in row-major order; the last wave, which gets the fewest, also owns
// the burden tile), two operand buffers in LDS, one barrier per iteration of 128 samples.  The operands are five times those
// of the hard-call kernel (5 bytes per genotype), which is why an iteration is 128 samples, not 256: 104 KB of LDS for
// M in 65..80.  Outputs as gene_suffstat_lat writes them (GeneDesc::hc == 2 with lat_den = 2^37): the integer K'K in the
// G'G block, true values in the [X | res] columns, the integer column sums; gene_assemble divides once.
#pragma once
#include "suffstat_hcx.hip.h"

namespace rvt {

constexpr int kFdxPlanes = 5;
constexpr int kFdxShift = 37;            // K = g 2^37
constexpr double kFdxMagic = 32768.0;    // 2^(52 - 37): g + kFdxMagic has ulp 2^-37
constexpr int kFdxNW = 4;                // loader waves = slices per iteration
constexpr int kFdxTW = 4;                // tile waves: a workgroup is 8 waves, two per SIMD (at most 256 registers each)
constexpr int kFdxWaveSteps = 2;         // 16-sample steps per loader wave and iteration (32 samples: one int8 operand)
constexpr int kFdxIterSteps = kFdxNW * kFdxWaveSteps;
constexpr int kFdxMaxMT = 5;
constexpr int kFdxEngineMT = 4;          // widest class the ENGINE sends here: M in 65..80 (five pairs of nine tiles per wave) spills — fp64 kernel
constexpr int kFdxStageCols = 16;        // null columns the kernel stages (2 d + 3 <= 16)
constexpr int kFdxMaxSteps = 512;        // steps per wave-part: 8 192 samples keep the order sums below 2^31
constexpr int kFdxOrders = 2 * kFdxPlanes - 1;

// Null-model operands (rvt_set_null): five balanced base-256 digit planes of the columns
//     [X_0 .. X_{d-1} | res | 1 | lo(X_0) .. lo(X_{d-1}) | lo(res)]
// Every null column x is the 46-bit fixed-point value 256 hi + lo (hi in five digits, lo ONE digit in [-128, 127] that sits in
// plane 0 of a column of its own): 2^-47 of the column's largest entry per entry, the precision the other integer kernels
// give their null tiles (five digits alone leave scores that differ from the fp64 path's in the twelfth digit).  The kernel
// adds the two tile columns of a null column when it stores them (lane v and lane v + d + 2 of a 16-lane row).  The column
// of ones is the integer 1 with scale 1: its tile column is the exact column sum of K.  Layout, in the order the tile waves
// read it: [group of 32 samples][plane 0..4][q 0..3][column k < ncols] x 8 bytes, byte 4 T + l = digit of sample
// 32 g + 16 T + 4 q + l; padded by eight groups.  value of column k = integer x scale[k] (a power of two).
struct NullTileF {
  const unsigned char* xq;
  double scale[16];
  int ncols;  // 2 d + 3 <= kFdxStageCols
};

// ---- tile pairs: (r, c) with r <= c <= MT in row-major order; c == MT: the null tile --------------------------------
constexpr int fdx_npairs(int MT) { return MT * (MT + 1) / 2 + MT; }
constexpr int fdx_per_wave(int MT) { return (fdx_npairs(MT) + kFdxTW - 1) / kFdxTW; }
struct FdxPair {
  int r, c;
};
constexpr FdxPair fdx_pair(int MT, int idx) {
  int r = 0;
  while (idx >= MT + 1 - r) {
    idx -= MT + 1 - r;
    ++r;
  }
  return FdxPair{r, r + idx};
}

// ---- LDS (bytes) -----------------------------------------------------------------------------------------------------
//   2 buffers x 4 slices x (5 MT + 1) x 512: plane p of column tile c at (5 c + p) 512, lane (v, q) 8 bytes (byte 4 T + l);
//        the burden operand last (lanes v = 0: c_cmc, v = 1: c_zeg; the other lanes zero)
//   xq stage: 2 buffers x 4 groups x 5 planes x 4 x kFdxStageCols x 8
//   tail: min / max bit patterns [2][MT 16] (64-bit), flag, count, sum of n^2
constexpr int fdx_slice_bytes(int MT) { return (kFdxPlanes * MT + 1) * 512; }
constexpr int fdx_buf_bytes(int MT) { return 2 * kFdxNW * fdx_slice_bytes(MT); }
constexpr int kFdxGroupMax = kFdxPlanes * 4 * kFdxStageCols * 8;
constexpr int kFdxStageBuf = kFdxNW * kFdxGroupMax;
constexpr int fdx_off_stage(int MT) { return fdx_buf_bytes(MT); }
constexpr int fdx_off_tail(int MT) { return fdx_off_stage(MT) + 2 * kFdxStageBuf; }
constexpr int fdx_tail_bytes(int MT) { return 2 * MT * 128 + 32; }
constexpr int fdx_lds_bytes(int MT) { return fdx_off_tail(MT) + fdx_tail_bytes(MT); }

typedef long fdx_op_t;  // the 8-byte operand of v_mfma_i32_16x16x32_i8

// One tile row of one step (four doubles of the lane's column): grid and range tests, the five digit dwords (byte l = digit
// of sample l), byte sums, min / max, burden hits.  `valid` = 0 zeroes the values (samples that do not exist).
//   bad    OR of (t - magic != g) over everything the lane has met (as a bit mask of compares)
//   hmax   max of the high dwords (unsigned: a negative value shows as a huge one)
//   C      the burden threshold of the column as a bit pattern: counted when bits(g) >= C, inverted for a flipped column
struct FdxCol {
  double mn, mx;
};
template <bool MASKED>
__device__ __forceinline__ void fdx_row(u4_t glo, u4_t ghi, unsigned (&dg)[kFdxPlanes], FdxCol& st, unsigned& bad, unsigned& hmax,
                                        unsigned long long C, unsigned fxor, unsigned& h, unsigned vmask) {
  if (MASKED) {
    // (entries that do not exist read as 0.0: digits 0, no burden hit; they take no part in min / max)
    glo[0] = (vmask & 0xffu) ? glo[0] : 0u;
    glo[1] = (vmask & 0xffu) ? glo[1] : 0u;
    glo[2] = (vmask & 0xff00u) ? glo[2] : 0u;
    glo[3] = (vmask & 0xff00u) ? glo[3] : 0u;
    ghi[0] = (vmask & 0xff0000u) ? ghi[0] : 0u;
    ghi[1] = (vmask & 0xff0000u) ? ghi[1] : 0u;
    ghi[2] = (vmask & 0xff000000u) ? ghi[2] : 0u;
    ghi[3] = (vmask & 0xff000000u) ? ghi[3] : 0u;
  }
  const unsigned lo[4] = {glo[0], glo[2], ghi[0], ghi[2]}, hi[4] = {glo[1], glo[3], ghi[1], ghi[3]};
  unsigned kl[4], kh[4], hit = 0u;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const double g = hc_dbl(lo[l], hi[l]);
    const double t = g + kFdxMagic;
    bad |= (t - kFdxMagic != g) ? 1u : 0u;
    hmax = hmax > hi[l] ? hmax : hi[l];
    if (!MASKED || ((vmask >> (8 * l)) & 0xffu)) {
      st.mn = __builtin_fmin(st.mn, g);
      st.mx = __builtin_fmax(st.mx, g);
    }
    const unsigned long long tb = __builtin_bit_cast(unsigned long long, t);
    // K + 0x8080808080: the low dword with its carry, then bits 32..38 of K, the bias byte and the carry
    const unsigned long long kb = (tb & 0x0000007FFFFFFFFFull) + 0x0000008080808080ull;
    kl[l] = (unsigned)kb;
    kh[l] = (unsigned)(kb >> 32);
    const unsigned long long gb = ((unsigned long long)hi[l] << 32) | lo[l];
    hit |= (gb >= C ? 1u : 0u) << (8 * l);
  }
  h += hit ^ fxor;
  // byte p of sample l -> byte l of plane p: a 4 x 4 byte transpose of the low dwords, the low bytes of the high ones
  const unsigned a0 = __builtin_amdgcn_perm(kl[1], kl[0], 0x05010400u), a1 = __builtin_amdgcn_perm(kl[1], kl[0], 0x07030602u);
  const unsigned b0 = __builtin_amdgcn_perm(kl[3], kl[2], 0x05010400u), b1 = __builtin_amdgcn_perm(kl[3], kl[2], 0x07030602u);
  unsigned u[kFdxPlanes];
  u[0] = __builtin_amdgcn_perm(b0, a0, 0x05040100u);
  u[1] = __builtin_amdgcn_perm(b0, a0, 0x07060302u);
  u[2] = __builtin_amdgcn_perm(b1, a1, 0x05040100u);
  u[3] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
  u[4] = __builtin_amdgcn_perm(kh[1], kh[0], 0x0c0c0400u) | __builtin_amdgcn_perm(kh[3], kh[2], 0x04000c0cu);
#pragma unroll
  for (int p = 0; p < kFdxPlanes; ++p) dg[p] = u[p] ^ 0x80808080u;
}

// the burden operand of one step: lane v = 0: c_cmc = (n > 0), v = 1: c_zeg = n (n <= 80), the other lanes zero; counts and the
// sum of n^2 stay with the loader (integers)
__device__ __forceinline__ unsigned fdx_burden_bytes(unsigned h, int v, unsigned& cnt, unsigned& zz) {
  const unsigned cc = ((h + 0x7f7f7f7fu) >> 7) & 0x01010101u;
  cnt += (unsigned)__builtin_popcount(cc);
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const unsigned n = (h >> (8 * l)) & 0xffu;
    zz = n * n + zz;
  }
  return v == 0 ? cc : (v == 1 ? h : 0u);
}

template <int MT>
__device__ __forceinline__ void suffstat_fdx_body(const GeneDesc& gd, const NullTileF& nt, long long N, long long ld, int d,
                                                  char* lds) {
  constexpr int NP = fdx_per_wave(MT), P = fdx_npairs(MT);
  constexpr int RING = 2;  // steps in flight per loader wave (one iteration ahead)
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int v = lane & 15, q = lane >> 4;
  const int wpart = blockIdx.x;
  if (wpart >= gd.n_wparts) return;
  const long long nsteps = ld >> 4;
  const long long s_begin = (long long)wpart * gd.steps_per_wpart;
  long long s_end = s_begin + gd.steps_per_wpart;
  if (s_end > nsteps) s_end = nsteps;
  if (s_begin >= s_end) return;
  const int M = gd.M;
  unsigned long long* const mm_all = reinterpret_cast<unsigned long long*>(lds + fdx_off_tail(MT));  // [2][MT 16]
  unsigned* const flagw = reinterpret_cast<unsigned*>(mm_all + 2 * MT * 16);  // [0] flags, [1] count, [2] sum of n^2
  char* const stage = lds + fdx_off_stage(MT);
  for (int x = threadIdx.x; x < 2 * MT * 16; x += (kFdxNW + kFdxTW) * 64) mm_all[x] = (x < MT * 16) ? ~0ull : 0ull;
  if (threadIdx.x < 8) flagw[threadIdx.x] = 0u;
  // the burden operand's lanes v >= 2 stay zero
  for (int x = threadIdx.x; x < 2 * kFdxNW * 128; x += (kFdxNW + kFdxTW) * 64)
    reinterpret_cast<unsigned*>(lds + (x / 128) * fdx_slice_bytes(MT) + kFdxPlanes * MT * 512)[x % 128] = 0u;
  __syncthreads();

#ifdef FDX_PROF
  long long prof[4] = {0, 0, 0, 0};  // cycles: [0] work, [1] waiting (vmcnt + barrier)
#define FDX_NOW() ((long long)__builtin_readcyclecounter())
#define FDX_TICK(k, t0) prof[k] += FDX_NOW() - (t0)
#define FDX_DUMP()                                                                                                       \
  if (gd.dbg_cmc && lane == 0)                                                                                           \
    for (int k = 0; k < 4; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(gd.dbg_cmc) + w * 4 + k, (unsigned long long)prof[k])
#else
#define FDX_NOW() 0ll
#define FDX_TICK(k, t0)
#define FDX_DUMP()
#endif
  const long long full = N >> 4;
  const long long s_fast_end = (s_end < full) ? s_end : full;
  const long long n_fast = (s_fast_end > s_begin) ? (s_fast_end - s_begin) / kFdxIterSteps : 0;
  const long long n_iter = (s_end - s_begin + kFdxIterSteps - 1) / kFdxIterSteps;
  double* out = gd.parts + (long long)wpart * gd.Mp * gd.Cp;
  const int Cp = gd.Cp;
  const int ncols = nt.ncols;
  const int group_bytes = kFdxPlanes * 4 * ncols * 8;

  if (w >= kFdxNW) {
    // ================================================ tile waves ================================================
    // This wave's pairs: indices tw NP .. of the row-major list (wave-uniform: scalar registers).  The LAST tile wave never has
    // one (P <= (kFdxTW - 1) NP for every class) and takes the burden tile into the accumulators of its slot 0.
    const int tw = w - kFdxNW;
    constexpr bool kOwnBurdenAcc = P > (kFdxTW - 1) * NP;  // (else slot 0 of the last wave, which then has no pair, serves)
    int pr_r[NP], pr_c[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      int idx = tw * NP + i, r = 0;
      const bool has = idx < P;
      idx = has ? idx : 0;
      while (idx >= MT + 1 - r) {
        idx -= MT + 1 - r;
        ++r;
      }
      pr_r[i] = has ? r : -1;
      pr_c[i] = r + idx;
    }
    const bool burden_wave = tw == kFdxTW - 1;
    i4_t acc[NP][kFdxOrders];
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
      for (int s = 0; s < kFdxOrders; ++s) acc[i][s] = i4_t{0, 0, 0, 0};
    i4_t accb[kOwnBurdenAcc ? kFdxPlanes : 1];
#pragma unroll
    for (int p = 0; p < (kOwnBurdenAcc ? kFdxPlanes : 1); ++p) accb[p] = i4_t{0, 0, 0, 0};
    const int null_plane = 4 * ncols * 8;                      // bytes of one null plane of a group
    const int xoff = (v < ncols) ? (q * ncols + v) * 8 : -1;   // the lane's entry of a null plane
    for (long long it = 0; it < n_iter; ++it) {
      const char* buf = lds + (int)(it & 1) * (kFdxNW * fdx_slice_bytes(MT));
      const char* xst = stage + (int)(it & 1) * kFdxStageBuf;
      const long long t_a = FDX_NOW();
      __syncthreads();
      FDX_TICK(1, t_a);
      const long long t_b = FDX_NOW();
      // Two 32-sample slices make ONE 64-sample operand of v_mfma_i32_16x16x64_i8 (bytes 0-7 from the even slice, 8-15 from the
      // odd one: any order of the samples along k serves, as long as both operands use it) — the 32-deep instruction costs
      // the same 16 cycles for half the samples.  The B operand of the next plane is fetched under the instructions of this one.
      auto op2 = [&](const char* p0, const char* p1) {
        const unsigned long long lo = *reinterpret_cast<const unsigned long long*>(p0), hi = *reinterpret_cast<const unsigned long long*>(p1);
        return i4_t{(int)(unsigned)lo, (int)(unsigned)(lo >> 32), (int)(unsigned)hi, (int)(unsigned)(hi >> 32)};
      };
      const i4_t zero4 = i4_t{0, 0, 0, 0};
#pragma unroll 1
      for (int ds = 0; ds < kFdxNW / 2; ++ds) {
        const char* s0 = buf + (2 * ds) * fdx_slice_bytes(MT) + lane * 8;
        const char* s1 = s0 + fdx_slice_bytes(MT);
        const char* x0 = xst + (2 * ds) * group_bytes + (xoff < 0 ? 0 : xoff);
        const char* x1 = x0 + group_bytes;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          if (pr_r[i] < 0) continue;  // (uniform)
          i4_t a[kFdxPlanes];
#pragma unroll
          for (int p = 0; p < kFdxPlanes; ++p) a[p] = op2(s0 + (kFdxPlanes * pr_r[i] + p) * 512, s1 + (kFdxPlanes * pr_r[i] + p) * 512);
          const bool null_tile = pr_c[i] == MT;
          const char* b0 = null_tile ? x0 : s0 + kFdxPlanes * pr_c[i] * 512;
          const char* b1 = null_tile ? x1 : s1 + kFdxPlanes * pr_c[i] * 512;
          const int bs = null_tile ? null_plane : 512;
          const bool bz = null_tile && xoff < 0;
          i4_t bn = op2(b0, b1);
#pragma unroll
          for (int qq = 0; qq < kFdxPlanes; ++qq) {
            i4_t b = bn;
            if (null_tile) b = bz ? zero4 : bn;  // (a uniform branch: Gram pairs pay no select)
            if (qq + 1 < kFdxPlanes) bn = op2(b0 + (qq + 1) * bs, b1 + (qq + 1) * bs);
#pragma unroll
            for (int p = 0; p < kFdxPlanes; ++p)
              acc[i][p + qq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[p], b, acc[i][p + qq], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (burden_wave) {
          const i4_t ab = op2(s0 + kFdxPlanes * MT * 512, s1 + kFdxPlanes * MT * 512);
#pragma unroll
          for (int p = 0; p < kFdxPlanes; ++p) {
            i4_t bx = op2(x0 + p * null_plane, x1 + p * null_plane);
            if (xoff < 0) bx = zero4;
            if (kOwnBurdenAcc)
              accb[kOwnBurdenAcc ? p : 0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ab, bx, accb[kOwnBurdenAcc ? p : 0], 0, 0, 0);
            else
              acc[0][p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ab, bx, acc[0][p], 0, 0, 0);
          }
        }
      }
      FDX_TICK(0, t_b);
    }
    FDX_DUMP();
    // ---- partial tiles: sum_s 256^s acc_s, from the top order down (i32 C/D map: lane (v, q), element i = row 4 q + i,
    // column v) ---------------------------------------------------------------------------------------------------
    auto value = [&](const i4_t (&a)[kFdxOrders], int i, int top) {
      double x = (double)a[top][i];
      for (int s = top - 1; s >= 0; --s) x = __builtin_fma(x, 256.0, (double)a[s][i]);
      return x;
    };
    double* cst0 = gd.colstat + (long long)wpart * kHcColstatRows * gd.Mp;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (pr_r[i] < 0) continue;
      if (pr_c[i] < MT) {
        const int col = pr_c[i] * 16 + v;
        if (col < M) {
#pragma unroll
          for (int e = 0; e < 4; ++e) out[(long long)(pr_r[i] * 16 + q * 4 + e) * Cp + col] = value(acc[i], e, kFdxOrders - 1);
        }
      } else {
        const double sc = nt.scale[v] * 0x1p-37;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = pr_r[i] * 16 + q * 4 + e;
          const double xi = value(acc[i], e, kFdxOrders - 1);
          const double x = xi * sc + __shfl_down(xi * sc, d + 2, 16);  // (+ the column's low digit: lane v + d + 2 of the row)
          if (M + v < Cp) out[(long long)row * Cp + M + v] = (v <= d) ? x : 0.0;
          if (M + 16 + v < Cp) out[(long long)row * Cp + M + 16 + v] = 0.0;
          if (v == d + 1) cst0[row] = xi;  // the ones column: the integer sum of K of the row's variant
        }
      }
    }
    __syncthreads();  // (the loaders' sums are in LDS)
    if (tw == kFdxTW - 1 && gd.bparts && lane < 16) {
      // ---- burden partial sums: [test][U, c'c, count, c'X_0 .. c'X_{d-1}], test 0 = CMC (tile row 0), 1 = Zeggini (row 1)
      auto planes = [&](int i) {
        if (!kOwnBurdenAcc) return value(acc[0], i, kFdxPlanes - 1);
        double x = (double)accb[kOwnBurdenAcc ? kFdxPlanes - 1 : 0][i];
#pragma unroll
        for (int p = kFdxPlanes - 2; p >= 0; --p) x = __builtin_fma(x, 256.0, (double)accb[kOwnBurdenAcc ? p : 0][i]);
        return x;
      };
      const double sc = nt.scale[v];
      double ac = planes(0) * sc, az = planes(1) * sc;
      ac += __shfl_down(ac, d + 2, 16);  // (+ the columns' low digits)
      az += __shfl_down(az, d + 2, 16);
      const int rl = 3 + d;
      double* bp = gd.bparts + (long long)wpart * 2 * rl;
      if (lane <= d) {
        const int k = (lane == d) ? 0 : 3 + lane;
        bp[k] = ac;
        bp[rl + k] = az;
      }
      if (lane == 0) {
        const double cn = (double)flagw[1];
        bp[1] = cn;
        bp[2] = cn;
        bp[rl + 1] = (double)flagw[2];
        bp[rl + 2] = cn;
      }
    }
    if (tw == 0) {
      // ---- column statistics (rows as suffstat_lat.hip.h writes them) ------------------------------------------------
      double* cst = gd.colstat + (long long)wpart * kHcColstatRows * gd.Mp;  // (row 0, the sums: by the null tiles' waves)
      for (int j = lane; j < MT * 16; j += 64) {
        const unsigned long long mnb = mm_all[j], mxb = mm_all[MT * 16 + j];
        const bool none = mnb == ~0ull;
        cst[gd.Mp + j] = none ? INFINITY : __builtin_bit_cast(double, mnb);
        cst[2 * gd.Mp + j] = none ? -INFINITY : __builtin_bit_cast(double, mxb);
        cst[3 * gd.Mp + j] = 0.0;
        reinterpret_cast<unsigned long long*>(cst)[4 * gd.Mp + j] = 0ull;
        reinterpret_cast<unsigned long long*>(cst)[5 * gd.Mp + j] = ~0ull;
      }
      if (gd.wflags && lane == 0) gd.wflags[wpart] = flagw[0] ? 2u : 0u;
    }
    return;
  }

  // ================================================== loader waves ==================================================
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
  const unsigned vfull = (unsigned)v * col_bytes + lane_off;
  const unsigned vlast = ((MT - 1) * 16 + v < M) ? (unsigned)((MT - 1) * 16 + v) * col_bytes + lane_off : 0x80000000u;
  const unsigned tile_bytes = (unsigned)__builtin_amdgcn_readfirstlane((int)(16u * col_bytes));
  auto gload = [&](int c, unsigned off_full, unsigned off_last, int imm) {
    return (c == MT - 1) ? __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, off_last + imm, 0, 0))
                         : __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, off_full + imm, c * tile_bytes, 0));
  };
  // burden threshold per column: (int)g' > 0 <=> g >= 1.0 for an unflipped column and NOT g > 1.0 for a flipped one
  int fxb = 0;
#pragma unroll
  for (int c = 0; c < MT; ++c) fxb |= (int)((gd.pflip[c] >> v) & 1) << c;
  FdxCol col[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    col[c].mn = INFINITY;
    col[c].mx = -INFINITY;
  }
  unsigned bad = 0u, hmax = 0u, cnt = 0u, zz = 0u;
  auto thr = [&](int c) { return ((fxb >> c) & 1) ? 0x3FF0000000000001ull : 0x3FF0000000000000ull; };
  auto fxo = [&](int c) { return ((fxb >> c) & 1) ? 0x01010101u : 0u; };
  auto fetch_operands = [&](long long i) {
    const long long g = ((s_begin + i * kFdxIterSteps) >> 1) + w;  // group of 32 samples
    hcx_dma(stage + (int)(i & 1) * kFdxStageBuf + w * group_bytes, nt.xq + g * group_bytes, group_bytes, lane);
  };
  auto put = [&](char* slice, int c, int u, const unsigned (&dg)[kFdxPlanes]) {
#pragma unroll
    for (int p = 0; p < kFdxPlanes; ++p) *reinterpret_cast<unsigned*>(slice + (kFdxPlanes * c + p) * 512 + lane * 8 + 4 * u) = dg[p];
  };
  long long it = 0;
  if (n_fast > 0) {
    unsigned offF = vfull + (unsigned)((s_begin + kFdxWaveSteps * w) * 128), offL = vlast + (unsigned)((s_begin + kFdxWaveSteps * w) * 128);
    u4_t glo[RING][MT], ghi[RING][MT];
    auto gload2 = [&](int c, int n_ahead, int half) {  // the wave's step n_ahead, counted from the current iteration's first
      const int imm = (n_ahead & 1) * 128 + half * 16;
      const unsigned so = (unsigned)((n_ahead >> 1) * kFdxIterSteps * 128);
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
    for (; it < n_fast; ++it) {
      const long long t_a = FDX_NOW();
      char* slice = lds + (int)(it & 1) * (kFdxNW * fdx_slice_bytes(MT)) + w * fdx_slice_bytes(MT);
      fetch_operands(it);
#pragma unroll
      for (int u = 0; u < kFdxWaveSteps; ++u) {
        unsigned h = 0;
#pragma unroll
        for (int c = 0; c < MT; ++c) {
          unsigned dg[kFdxPlanes];
          fdx_row<false>(glo[u][c], ghi[u][c], dg, col[c], bad, hmax, thr(c), fxo(c), h, 0xffffffffu);
          put(slice, c, u, dg);
          glo[u][c] = gload2(c, u + RING, 0);
          ghi[u][c] = gload2(c, u + RING, 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        h = row16_sum(h);
        const unsigned bb = fdx_burden_bytes(h, v, cnt, zz);
        if (v < 2) *reinterpret_cast<unsigned*>(slice + kFdxPlanes * MT * 512 + lane * 8 + 4 * u) = bb;
        __builtin_amdgcn_sched_barrier(0);
      }
      FDX_TICK(0, t_a);
      const long long t_b = FDX_NOW();
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING * MT * 2) : "memory");
      FDX_TICK(2, t_b);
      const long long t_c = FDX_NOW();
      __syncthreads();
      FDX_TICK(1, t_c);
      offF += kFdxIterSteps * 128;
      offL += kFdxIterSteps * 128;
    }
  }
  for (; it < n_iter; ++it) {  // ragged end: every step from a clamped position, masked
    const long long s0 = s_begin + it * kFdxIterSteps + kFdxWaveSteps * w;
    char* slice = lds + (int)(it & 1) * (kFdxNW * fdx_slice_bytes(MT)) + w * fdx_slice_bytes(MT);
    fetch_operands(it);
#pragma unroll 1
    for (int u = 0; u < kFdxWaveSteps; ++u) {
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
        unsigned dg[kFdxPlanes];
        fdx_row<true>(glo, ghi, dg, col[c], bad, hmax, thr(c), fxo(c) & vmask, h, vmask);
        put(slice, c, u, dg);
      }
      h = row16_sum(h) & vmask;
      const unsigned bb = fdx_burden_bytes(h, v, cnt, zz);
      if (v < 2) *reinterpret_cast<unsigned*>(slice + kFdxPlanes * MT * 512 + lane * 8 + 4 * u) = bb;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  FDX_DUMP();
  // ---- the loaders' sums, min / max and flags meet in LDS --------------------------------------------------------------
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    double mn = col[c].mn, mx = col[c].mx;
    mn = __builtin_fmin(mn, __shfl_xor(mn, 16, 64));
    mx = __builtin_fmax(mx, __shfl_xor(mx, 16, 64));
    mn = __builtin_fmin(mn, __shfl_xor(mn, 32, 64));
    mx = __builtin_fmax(mx, __shfl_xor(mx, 32, 64));
    if (lane < 16 && mn <= mx) {  // (values are >= 0 wherever the gene is not handed back: bit patterns order like values)
      __hip_atomic_fetch_min(mm_all + c * 16 + lane, __builtin_bit_cast(unsigned long long, mn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_max(mm_all + MT * 16 + c * 16 + lane, __builtin_bit_cast(unsigned long long, mx), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  {
    unsigned cn = (v == 0) ? cnt : 0u, z2 = (v == 0) ? zz : 0u;
    cn += __shfl_xor(cn, 16, 64);
    z2 += __shfl_xor(z2, 16, 64);
    cn += __shfl_xor(cn, 32, 64);
    z2 += __shfl_xor(z2, 32, 64);
    const bool off = __builtin_amdgcn_ballot_w64(bad != 0u || hmax > 0x40000000u) != 0ull;
    if (lane == 0) {
      __hip_atomic_fetch_add(flagw + 1, cn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(flagw + 2, z2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (off) __hip_atomic_fetch_or(flagw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
}

template <int TOP>
__global__ __launch_bounds__((kFdxNW + kFdxTW) * 64, 2) void gene_suffstat_fdx_any(const GeneDesc* __restrict__ genes, NullTileF nt,
                                                                             long long N, long long ld, int d) {
  const GeneDesc gd = genes[blockIdx.y];
  __shared__ __attribute__((aligned(16))) char lds[fdx_lds_bytes(TOP)];
  switch (gd.MT) {
    case 1: suffstat_fdx_body<1>(gd, nt, N, ld, d, lds); break;
    case 2: suffstat_fdx_body<2>(gd, nt, N, ld, d, lds); break;
    case 3: suffstat_fdx_body<3>(gd, nt, N, ld, d, lds); break;
    case 4: suffstat_fdx_body<4>(gd, nt, N, ld, d, lds); break;
    case 5:
      if constexpr (TOP >= 5) suffstat_fdx_body<5>(gd, nt, N, ld, d, lds);
      break;
    default: break;
  }
}
template <int MT>
__global__ __launch_bounds__((kFdxNW + kFdxTW) * 64, 2) void gene_suffstat_fdx(const GeneDesc* __restrict__ genes, NullTileF nt,
                                                                         long long N, long long ld, int d) {
  const GeneDesc gd = genes[blockIdx.y];
  if (gd.MT != MT) return;
  __shared__ __attribute__((aligned(16))) char lds[fdx_lds_bytes(MT)];
  suffstat_fdx_body<MT>(gd, nt, N, ld, d, lds);
}

}  // namespace rvt
