// rvtests_amd — sufficient statistics of a gene straight from its PLINK 2-bit rows (rvt_submit_gene_bed), unweighted
// (quantitative-trait) null model.
//
// The packed hand-off used to be EXPANDED on the device into the boundary's fp64 block (consolidate_write_kernel: 200 MB
// written per gene at N = 500 000) which gene_suffstat_hc then read back — 400 MB of HBM traffic for 6 MB of information,
// and what held the 2-bit feed at 4.5 k gene-sets/s where the link carries 8 k (DESIGN.md "Host feed").  Here the rows are
// read as they arrived: one 16-byte load per column and 64 samples, i.e. 1/64 of the bytes, same arithmetic:
//   * a 2-bit code (libVcf/PlinkInputFile.h:206-209: 00 -> 0, 10 -> 1, 11 -> 2, 01 -> missing) becomes the integer H and
//     the mask m of suffstat_hc.hip.h's masked-entry scheme by byte-parallel logic; the operand byte is H + 4 m, the
//     ordinary tiles hold C = (H + 4m)'(H + 4m), the tiles P' and Q are added to the 16-bit LDS counters when a 64-sample
//     operand holds a missing call (hc_group_end), and gene_assemble recovers G'G exactly as for an fp64 block whose
//     missing calls imputeGenotypeToMean has filled;
//   * the value imputeGenotypeToMean would have written — mu_j per column, from the count pass over the packed rows
//     (consolidate_count_kernel<bed2_t> / consolidate_fill_kernel: the reference's truncating allele count) — is known
//     BEFORE this kernel runs (header of the packed block), so G'[X | rr] is formed with the true values on the fp64 matrix
//     cores, and the burden collapse is exact in the same pass: the flip of a column (sum > N), whether it is
//     polymorphic, and whether its imputed value counts ((int)mu' > 0) come from the header — no prediction, no fallback;
//   * outputs as gene_suffstat_hc writes them (partial tiles, six-row column statistics with the bit pattern of mu as
//     the OR / AND rows, packed P' / Q images, burden partial sums): everything behind this kernel is unchanged.
// Bound by the vector and matrix pipes, not by memory (≈6 MB per gene).
#pragma once
#include "suffstat_hc.hip.h"

namespace rvt {

constexpr int kHcpHeaderBytes = 1024;  // in front of the packed rows
struct HcpHeader {                     // written by hcp_header_kernel (fam_kernels.hip.h) after the count pass
  double mu[96];                       // imputed value of column j (0 when the column has no missing call)
  unsigned short flip[8], poly[8], cm[8];  // per 16-column block: sum > N; min != max; the imputed value counts
};
static_assert(sizeof(HcpHeader) <= kHcpHeaderBytes, "header does not fit");

// codes of four samples (one byte of a PLINK row) -> H (0 / 1 / 2 per byte; 0 where missing) and m (1 where missing)
__device__ __forceinline__ void hcp_decode(unsigned b, unsigned& p, unsigned& m) {
  const unsigned w = (b | (b << 6) | (b << 12) | (b << 18)) & 0x03030303u;  // one 2-bit code per byte
  const unsigned hi = (w >> 1) & 0x01010101u, lo = w & 0x01010101u;
  m = lo & ~hi;         // 01
  p = hi + (hi & lo);   // 10 -> 1, 11 -> 2
}

// the double a sample's (H, m) stands for: 0.0 / 1.0 / 2.0, or the column's imputed value
__device__ __forceinline__ double hcp_value(unsigned p, unsigned m, int l, unsigned mu_lo, unsigned mu_hi) {
  const unsigned H = (p >> (8 * l)) & 3u;
  unsigned hi = H ? 0x3FE00000u + (H << 20) : 0u;  // 0x3FF00000, 0x40000000
  unsigned lo = 0u;
  if ((m >> (8 * l)) & 1u) {
    hi = mu_hi;
    lo = mu_lo;
  }
  return hc_dbl(lo, hi);
}

struct HcpCol {  // per lane and tile row
  unsigned fx, pm, cmk, mu_lo, mu_hi;
};

// one tile row of one step.  b = the byte with the lane's four samples; vb = 0xff in the bytes of samples that exist
// (WITH_T = false: G'[X | rr] is formed by gene_tnull_hcp from digit planes of the null tile, see below)
template <bool WITH_T>
__device__ __forceinline__ void hcp_row(unsigned b, const double (&xv)[4], d4_t& accT, unsigned& pk, unsigned& cs,
                                        const HcpCol& cl, unsigned& h, unsigned& anym, unsigned vb) {
  unsigned p, m;
  hcp_decode(b, p, m);
  p &= vb;
  m &= vb;
  if constexpr (WITH_T) {
#pragma unroll
    for (int l = 0; l < 4; ++l)
      accT = __builtin_amdgcn_mfma_f64_16x16x4f64(hcp_value(p, m, l, cl.mu_lo, cl.mu_hi), xv[l], accT, 0, 0, 0);
  }
  cs = __builtin_amdgcn_sad_u8(p, 0u, cs);
  const unsigned t = p ^ cl.fx;  // flipped column: (int)(2 - g) > 0  <=>  g != 2
  h += ((((t | (t >> 1)) & 0x01010101u) & ~m) | (m & cl.cmk)) & cl.pm;
  pk = p | (m << 2);  // operand byte H + 4 m
  anym |= m;
}

template <int MT, bool WITH_T>
__device__ __forceinline__ void suffstat_hcp_body(const GeneDesc& gd, const NullTile& nt, long long N, long long ld,
                                                  int d, unsigned* lds) {
  const int lane = threadIdx.x & 63;
  const int v = lane & 15, q = lane >> 4;
  const int wpart = blockIdx.x;
  if (wpart >= gd.n_wparts) return;
  constexpr int kPQ = hc_pq_words(MT);
#pragma unroll 4
  for (int w = lane; w < kPQ; w += 64) lds[w] = 0u;
#pragma unroll
  for (int c = 0; c < MT; ++c) lds[kPQ + 64 * c + lane] = (lane & 2) ? 0xffffffffu : 0u;
  if (lane < 4) lds[kPQ + 64 * MT + lane] = 0u;
  unsigned anym = 0u;
  const long long nsteps = ld >> 4;
  const long long s_begin = (long long)wpart * gd.steps_per_wpart;  // (a multiple of 4 steps: kHcStepUnit)
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
  const unsigned char* blk = reinterpret_cast<const unsigned char*>(uniform(gd.G));
  const HcpHeader* hdr = reinterpret_cast<const HcpHeader*>(blk);
  const unsigned pitch = (unsigned)gd.pk_pitch;
  const __amdgpu_buffer_rsrc_t rp =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(blk + kHcpHeaderBytes), 0, (unsigned)M * pitch, 0x00020000);
  const unsigned xbytes = (unsigned)((unsigned long long)nt.cols * (unsigned long long)ld * 8ull);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(uniform(nt.base), 0, xbytes, 0x00020000);
  const unsigned col_bytes = (unsigned)((unsigned long long)ld * 8ull);
  const int xcol = (v <= d) ? v : d + 1;  // X_k, rr, or the zero column
  const unsigned xbase = (unsigned)xcol * col_bytes + (unsigned)(q * 32);
  unsigned cbase[MT];
  HcpCol cl[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    const int col = c * 16 + v;
    const bool in = col < M;
    cbase[c] = in ? (unsigned)col * pitch : 0x80000000u;  // a pad column reads zeros
    const double mu = in ? hdr->mu[col] : 0.0;
    const unsigned long long mb = __builtin_bit_cast(unsigned long long, mu);
    cl[c].mu_lo = (unsigned)mb;
    cl[c].mu_hi = (unsigned)(mb >> 32);
    cl[c].fx = (in && ((hdr->flip[c] >> v) & 1)) ? 0x02020202u : 0u;
    cl[c].pm = (in && ((hdr->poly[c] >> v) & 1)) ? 0x01010101u : 0u;
    cl[c].cmk = (in && ((hdr->cm[c] >> v) & 1)) ? 0x01010101u : 0u;
  }
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

  // groups of 4 steps = 64 samples = 16 bytes of every row.  A group that reaches beyond N (or beyond the wave's range) is
  // masked sample by sample; rows are padded to 16 bytes and loads beyond the block return zeros.
  for (long long s = s_begin; s < s_end; s += 4) {
    u4_t cw[MT];
#pragma unroll
    for (int c = 0; c < MT; ++c)
      cw[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rp, cbase[c] + (unsigned)(s * 4), 0, 0));
    u4_t xlo[4], xhi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long su = (s + u < s_end) ? s + u : s_end - 1;  // (clamped: its samples are masked below)
      xlo[u] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xbase + (unsigned)(su * 128), 0, 0));
      xhi[u] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, xbase + (unsigned)(su * 128) + 16, 0, 0));
    }
    const bool whole = (s + 4 <= s_end) && ((s + 4) * 16 <= N);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      unsigned vmask = 0xffffffffu;
      if (!whole) {
        vmask = 0u;
        const long long smp = (s + u) * 16 + q * 4;
#pragma unroll
        for (int l = 0; l < 4; ++l) vmask |= (s + u < s_end && smp + l < N) ? (0xffu << (8 * l)) : 0u;
      }
      double xv[4] = {hc_dbl(xlo[u][0], xlo[u][1]), hc_dbl(xlo[u][2], xlo[u][3]), hc_dbl(xhi[u][0], xhi[u][1]),
                      hc_dbl(xhi[u][2], xhi[u][3])};
      if (!whole) {
#pragma unroll
        for (int l = 0; l < 4; ++l) xv[l] = ((vmask >> (8 * l)) & 1u) ? xv[l] : 0.0;
      }
      unsigned h = 0;
#pragma unroll
      for (int c = 0; c < MT; ++c) {
        const unsigned b = (cw[c][u] >> (8 * q)) & 0xffu;
        hcp_row<WITH_T>(b, xv, accT[c], pk[c][u], cs[c], cl[c], h, anym, vmask);
      }
      if (whole)
        hc_finish<false>(h, xv, bu, 0xffffffffu);
      else
        hc_finish<true>(h, xv, bu, vmask);
    }
    hc_group_end<MT>(pk, accS, anym, lds, lane);
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
      if (WITH_T && M + v < Cp) out[(long long)row * Cp + M + v] = accT[r][i];  // (else: written by gene_tnull_hcp)
      if (M + 16 + v < Cp) out[(long long)row * Cp + M + 16 + v] = 0.0;
    }
  }
  // ---- column statistics: sum of H, min / max over the hard calls, masked count, the bit pattern of mu ------------------
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
      const i4_t dg = accS[t];
      const int sel = (lane & 3) == 0 ? dg[0] : ((lane & 3) == 1 ? dg[1] : ((lane & 3) == 2 ? dg[2] : dg[3]));
      const int diag = __shfl(sel, v + 16 * (v >> 2), 64);
      const int tq = MT * MT + t;
      const unsigned qw = lds[(tq * 2 + ((v >> 1) & 1)) * 64 + v + 16 * (v >> 2)];
      const long long nm = (long long)((qw >> (16 * (v & 1))) & 0xffffu);  // missing calls of column c * 16 + v
      t += MT - c;
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
        const unsigned long long mb = ((unsigned long long)cl[c].mu_hi << 32) | cl[c].mu_lo;
        unsigned long long* bits = reinterpret_cast<unsigned long long*>(cst);
        bits[4 * gd.Mp + j] = nm > 0 ? mb : 0ull;
        bits[5 * gd.Mp + j] = nm > 0 ? mb : ~0ull;
      }
    }
  }
  if (gd.wflags && lane == 0) gd.wflags[wpart] = wflag & 1u;
  if ((wflag & 1u) && gd.pq) {
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

template <int MT, int WAVES, bool WITH_T = true>
__global__ __launch_bounds__(64, WAVES) void gene_suffstat_hcp(const GeneDesc* __restrict__ genes, NullTile nt, long long N,
                                                               long long ld, int d) {
  __shared__ unsigned lds[hc_lds_words(MT)];
  const GeneDesc gd = genes[blockIdx.y];
  if (gd.MT != MT) return;
  suffstat_hcp_body<MT, WITH_T>(gd, nt, N, ld, d, lds);
}

// ---- G'[X | rr] of a packed gene on the int8 matrix cores (round 6) --------------------------------------------------------------
// Two thirds of gene_suffstat_hcp went into this product: one fp64 matrix instruction per four samples and tile row whose A
// operand — the double a 2-bit code stands for — costs six vector instructions per genotype (35 us of chip time per gene of
// M = 50 at N = 500 000; 11.5 without it).  The hard calls ARE small integers, so the product belongs on the int8 instruction:
// the null tile [X_0 .. X_{d-1} | rr] is quantised once per null model to EIGHT balanced base-128 digit planes per column of
// its 56-bit fixed-point values (a power-of-two scale per column; rvt_set_null, as the six planes of the weighted kernel,
// suffstat_hcx.hip.h) and stored in operand order:
//   xq[(((g 8 + p) 4 + q) ncx + k) 16 + 4 T + l] = digit p of column k at sample 64 g + 16 T + 4 q + l
// — the byte order of the operand gene_suffstat_hcp builds from a row's 16 bytes.  Per group of 64 samples and tile row: four
// paired instructions (hcx_pair_step: 128 (A'B_2j) + A'B_2j+1, exact in int32 over a wave-part of at most 131 072 samples)
// with A = H, the integer part of the codes (0 where the call is missing).  Mean-imputed columns: g = H + mu m, so a second pass
// over the wave-part — only when it holds a missing call — forms m'[X | rr] the same way and the result is H'X + mu m'X.
// Every sum is an exact integer of the quantised tile; the quantisation is 2^-56 of twice a column's largest entry per
// sample (relative 1e-13 on these sums: below the rounding of the fp64 product it replaces).  Writes the T columns of the
// wave-part's partial image (out[row Cp + M + k]) that gene_suffstat_hcp<.., false> leaves alone.
constexpr int kHcpPlanes = 8, kHcpPairs = 4;
constexpr long long kHcpPlaneMaxSamples = 131072;  // (int32 range of a pair sum: 64 x 2 x 63 x 129 per group)
struct HcpPlanes {
  const unsigned char* xq;
  const double* scale;  // (device, 16 entries) value of column k = integer x scale[k]
  int ncx;              // d + 1 columns: X_0 .. X_{d-1}, rr
};
// two digit planes of one tile and one 64-sample operand: acc = 128 (A'B0) + A'B1 + acc, exactly, in int32 (as hcx_pair_step)
__device__ __forceinline__ void hcp_pair_step(i4_t& acc, const i4_t& a, const i4_t& b0, const i4_t& b1) {
  const i4_t z = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b0, i4_t{0, 0, 0, 0}, 0, 0, 0);
  i4_t t = acc;
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = (int)(((unsigned)z[i] << 7) + (unsigned)t[i]);
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b1, t, 0, 0, 0);
}

// (STATS: also the column statistics gene_suffstat_hcp keeps — sum of H, sum of H^2, number of missing calls per column and
//  lane — for callers that need nothing else of that kernel, see gene_tnull_hcp<MT, true>)
template <int MT, bool MASK, bool STATS = false>
__device__ __forceinline__ bool hcp_tnull_pass(i4_t (&acc)[kHcpPairs][MT], const __amdgpu_buffer_rsrc_t& rp, const unsigned (&cbase)[MT],
                                               const unsigned char* xq_lane, long long plane_stride, bool has_col,
                                               long long s_begin, long long s_end, int q, unsigned (*st)[3] = nullptr) {
  unsigned anym = 0u;
  for (long long s = s_begin; s < s_end; s += 4) {
    u4_t cw[MT];
#pragma unroll
    for (int c = 0; c < MT; ++c)
      cw[c] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rp, cbase[c] + (unsigned)(s * 4), 0, 0));
    i4_t b[kHcpPlanes];
    const unsigned char* xg = xq_lane + (s >> 2) * (kHcpPlanes * plane_stride);
#pragma unroll
    for (int p = 0; p < kHcpPlanes; ++p) {
      u4_t t = u4_t{0u, 0u, 0u, 0u};
      if (has_col) t = *reinterpret_cast<const u4_t*>(xg + p * plane_stride);
      b[p] = i4_t{(int)t[0], (int)t[1], (int)t[2], (int)t[3]};
    }
#pragma unroll
    for (int c = 0; c < MT; ++c) {
      i4_t a;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        unsigned pv, mv;
        hcp_decode((cw[c][u] >> (8 * q)) & 0xffu, pv, mv);
        if (s + u >= s_end) pv = mv = 0u;
        anym |= mv;
        a[u] = (int)(MASK ? mv : pv);
        if constexpr (STATS) {
          st[c][0] = __builtin_amdgcn_sad_u8(pv, 0u, st[c][0]);                                        // sum H
          st[c][1] = __builtin_amdgcn_sad_u8(pv + (((pv >> 1) & 0x01010101u) << 1), 0u, st[c][1]);    // sum H^2 (2 -> 4)
          st[c][2] = __builtin_amdgcn_sad_u8(mv, 0u, st[c][2]);                                        // missing calls
        }
      }
#pragma unroll
      for (int j = 0; j < kHcpPairs; ++j) hcp_pair_step(acc[j][c], a, b[2 * j], b[2 * j + 1]);
    }
  }
  return __any(anym != 0u);
}

// SCORE = true: the single-variant score test needs of a slice only its columns' T rows, the diagonal of its Gram tile and the
// column statistics (score_finish_kernel) — all of which this kernel has in hand: sum H^2 + 16 nm is the diagonal entry of the
// operand's Gram tile, the statistics rows are those gene_suffstat_hcp writes.  gene_suffstat_hcp is then not launched at all.
template <int MT, bool SCORE = false>
__global__ __launch_bounds__(64) void gene_tnull_hcp(const GeneDesc* __restrict__ genes, HcpPlanes pl, long long N, long long ld) {
  const GeneDesc gd = genes[blockIdx.y];
  if (gd.MT != MT) return;
  const int lane = threadIdx.x & 63, v = lane & 15, q = lane >> 4;
  const int wpart = blockIdx.x;
  if (wpart >= gd.n_wparts) return;
  const long long nsteps = ld >> 4;
  const long long s_begin = (long long)wpart * gd.steps_per_wpart;
  long long s_end = s_begin + gd.steps_per_wpart;
  if (s_end > nsteps) s_end = nsteps;
  if (s_begin >= s_end) return;
  const int M = gd.M;
  const unsigned char* blk = reinterpret_cast<const unsigned char*>(gd.G);
  const HcpHeader* hdr = reinterpret_cast<const HcpHeader*>(blk);
  const unsigned pitch = (unsigned)gd.pk_pitch;
  const __amdgpu_buffer_rsrc_t rp =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(blk + kHcpHeaderBytes), 0, (unsigned)M * pitch, 0x00020000);
  unsigned cbase[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c) cbase[c] = (c * 16 + v < M) ? (unsigned)(c * 16 + v) * pitch : 0x80000000u;  // pad column: zeros
  const bool has_col = v < pl.ncx;
  const long long plane_stride = 4ll * pl.ncx * 16;  // bytes of one plane of one group: [q][k] x 16
  const unsigned char* xq_lane = pl.xq + ((long long)q * pl.ncx + (has_col ? v : 0)) * 16;
  const double sc = has_col ? pl.scale[v] : 0.0;
  i4_t acc[kHcpPairs][MT];
  auto clear = [&]() {
#pragma unroll
    for (int j = 0; j < kHcpPairs; ++j)
#pragma unroll
      for (int c = 0; c < MT; ++c) acc[j][c] = i4_t{0, 0, 0, 0};
  };
  // the integer the four pair sums encode, times the column's scale: sum_j pair_j 128^(6 - 2 j)
  auto value = [&](int c, int i) {
    const double hi = (double)acc[0][c][i] * 0x1p42 + (double)acc[1][c][i] * 0x1p28;
    const double lo = (double)acc[2][c][i] * 0x1p14 + (double)acc[3][c][i];
    return (hi + lo) * sc;
  };
  clear();
  unsigned st[MT][3];
#pragma unroll
  for (int c = 0; c < MT; ++c) st[c][0] = st[c][1] = st[c][2] = 0u;
  const bool masked = hcp_tnull_pass<MT, false, SCORE>(acc, rp, cbase, xq_lane, plane_stride, has_col, s_begin, s_end, q, st);
  double t[MT][4];
#pragma unroll
  for (int c = 0; c < MT; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) t[c][i] = value(c, i);
  if (masked) {  // (wave-uniform)
    clear();
    (void)hcp_tnull_pass<MT, true>(acc, rp, cbase, xq_lane, plane_stride, has_col, s_begin, s_end, q);
#pragma unroll
    for (int c = 0; c < MT; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = c * 16 + q * 4 + i;
        t[c][i] += (row < M ? hdr->mu[row] : 0.0) * value(c, i);
      }
  }
  double* out = gd.parts + (long long)wpart * gd.Mp * gd.Cp;
  const int Cp = gd.Cp;
  if (M + v < Cp) {
#pragma unroll
    for (int c = 0; c < MT; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[(long long)(c * 16 + q * 4 + i) * Cp + M + v] = t[c][i];  // (i32 C/D map: row 4 q + i, column v)
  }
  if constexpr (SCORE) {
    // (what gene_suffstat_hcp leaves for gene_flags_hc_kernel: "no entry of this wave-part is outside the codes" — the word lives in
    //  a work space the batch before may have used, and a stale bit there hands the slice to the fp64 kernel, which would read
    //  the packed rows as doubles)
    if (gd.wflags && lane == 0) gd.wflags[wpart] = 0u;
    long long cnt_w = ((s_end * 16 < N) ? s_end * 16 : N) - s_begin * 16;
    if (cnt_w < 0) cnt_w = 0;
    double* cst = gd.colstat + (long long)wpart * kHcColstatRows * gd.Mp;
#pragma unroll
    for (int c = 0; c < MT; ++c) {
      unsigned sm = st[c][0], hh = st[c][1], nmu = st[c][2];
      sm += __shfl_xor(sm, 16, 64);
      hh += __shfl_xor(hh, 16, 64);
      nmu += __shfl_xor(nmu, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      hh += __shfl_xor(hh, 32, 64);
      nmu += __shfl_xor(nmu, 32, 64);
      if (lane < 16) {
        const int j = c * 16 + lane;
        const long long nm = (long long)nmu, n2 = ((long long)hh - (long long)sm) / 2, n1 = 2 * (long long)sm - (long long)hh,
                        n0 = cnt_w - n1 - n2 - nm;
        cst[j] = (double)sm;
        cst[gd.Mp + j] = n0 > 0 ? 0.0 : (n1 > 0 ? 1.0 : (n2 > 0 ? 2.0 : INFINITY));
        cst[2 * gd.Mp + j] = n2 > 0 ? 2.0 : (n1 > 0 ? 1.0 : (n0 > 0 ? 0.0 : -INFINITY));
        cst[3 * gd.Mp + j] = (double)nm;
        const unsigned long long mb = j < M ? __builtin_bit_cast(unsigned long long, hdr->mu[j]) : 0ull;
        unsigned long long* bits = reinterpret_cast<unsigned long long*>(cst);
        bits[4 * gd.Mp + j] = nm > 0 ? mb : 0ull;
        bits[5 * gd.Mp + j] = nm > 0 ? mb : ~0ull;
        if (j < M) out[(long long)j * Cp + j] = (double)((long long)hh + 16 * nm);  // the diagonal of (H + 4 m)'(H + 4 m)
      }
    }
  }
}

}  // namespace rvt
