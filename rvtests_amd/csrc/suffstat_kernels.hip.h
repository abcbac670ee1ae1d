// rvtests_amd — the sufficient-statistics kernels (templates only, so that several translation units can
// instantiate them: the engine is compiled as three objects in parallel — unweighted K2, weighted K2, everything else).
//
//  gene_suffstat_mfma<group,W>  fp64-MFMA sufficient statistics  R = G'·D·[G | X | rr]  + exact column
//                               sum/min/max + per-sample ">=1" / "<=1" bit masks, ONE pass over G
//  gene_suffstat_panel<W>       the same for genes wider than 96 variants, tile grid cut into 4 x 4 panels
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/rvtests_amd.h"

namespace rvt {

struct GeneStats;

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
// pointers read from descriptors in memory are generic; loads through them would be flat_load (which also ties
// up lgkmcnt and forces a full drain).  These are known to be global memory.
typedef const double __attribute__((address_space(1))) * gcdp_t;
typedef const d2_t __attribute__((address_space(1))) * gcd2p_t;
__device__ __forceinline__ gcdp_t as_global(const double* p) { return (gcdp_t)(unsigned long long)p; }

// One gene as the kernels see it.
struct GeneDesc {
  const double* G;       // N x M block, leading dimension ld
  int M, MT, CT;         // columns, row tiles (ceil(M/16)), column tiles (ceil((M+d+1)/16))
  int Mp, Cp;            // 16*MT, 16*CT
  int n_wparts;          // wave-parts the sample axis is cut into
  int steps_per_wpart;   // 16-sample steps per wave-part
  double* parts;         // n_wparts x Mp x Cp partial statistics (row-major)
  double* colstat;       // n_wparts x 3 x Mp (hard-call path: n_wparts x kHcColstatRows x Mp, suffstat_hc.hip.h)
  unsigned long long* masks;  // [2][nsteps][MT][4] ballots: kind 0 "g >= 1", kind 1 "g <= 1"
  unsigned short* flags;      // [2][MT]: flip bits, polymorphic bits per 16-variant block
  double* bparts;        // n_bparts x 2 x (3+d)
  double* scratch;       // gene_scratch_doubles(Mp, Cp)
  double* lambda;        // 2*M
  void* qags_mem;        // qags_workspace_bytes(1000)
  const double* af;      // M allele frequencies (device copy)
  GeneStats* stats;
  rvt_gene_result* result;
  int* dbg_flip;         // optional M ints
  int* dbg_kept;         // optional M ints
  double* dbg_cmc;       // optional N doubles
  double* dbg_zeg;       // optional N doubles
  long long gene_id;
  // hard-call path (suffstat_hc.hip.h)
  unsigned short pflip[8];  // predicted flip bits (af > 0.5) per 16-variant block, first 6 blocks
  int n_bparts;             // burden partial records of this gene (wave-parts on the hard-call path)
  int hcp_planes;           // hc == 3: G'[X | rr] from the digit planes of the null tile (gene_tnull_hcp; resident .bed genes)
  int hc;                   // 4: gene_suffstat_fdx (float-precision dosages: as 2, with lat_den = 2^37),
                            // 1: gene_suffstat_hc / _hcw (hard calls), 2: gene_suffstat_lat (lattice dosages), 3: gene_suffstat_hcp
                            // (PLINK 2-bit rows), 0: general kernel
  double* vt_mem;           // AnalyticVT workspace (gene_vt_doubles(Mp)), null unless the test is requested
  unsigned* pq;             // hard-call path: n_wparts x hc_pq_words(MT) packed 16-bit counters of the masked tiles
  unsigned* wflags;         // hard-call path: per wave-part, bit 0 = masked entries met (pq written), bit 1 = bad entry
  int pk_pitch;             // hc == 3 (gene_suffstat_hcp): G points to a packed block — header, then M rows of 2-bit codes,
                            // pk_pitch bytes apart
  double lat_den;           // hc == 2: the lattice denominator — the G'G tiles and column sums of `parts` / `colstat` are
                            // the INTEGERS K'K and sum K (exact through the reduction), divided once in gene_assemble
  unsigned long long* pqw;  // weighted hard-call path (suffstat_hcx.hip.h): the gene's masked-entry tables P = H'Vm (Mp x Mp) and
                            // Q = m'Vm (Mp x Mp, upper triangle) as 64-bit integers in units of 2^-42, zeroed by the host
  int pv_gene;              // gene_pvalue_kernel: workgroup b of the launch takes gene genes[b].pv_gene — the batch in order of
                            // falling M (longest p-value work first, so the short ones fill the tail of the launch)
};

struct NullDev {
  const double* X;     // ld x d  (column k at X + k*ld)
  const double* res;   // ld
  const double* rr;    // ld: res (quantitative) or res / v (binary)
  const double* v;     // ld
  const double* zeros; // ld zeros
};

// =====================================================================================================
// K2: sufficient statistics on the fp64 matrix cores.
//
// v_mfma_f64_16x16x4_f64 computes D(16x16) += A(16x4)·B(4x16) with lane l holding A[l&15][l>>4] and
// B[l>>4][l&15].  Here the contraction index is the SAMPLE, A's row index is a variant and B's column
// index is a column of [G | X | rr]; since A and B use the same (index, k) lane map, the register that
// holds 16 variants x 4 samples of G serves as the A operand of one tile row AND as the B operand of
// one tile column — every element of G is loaded from HBM exactly once, by exactly one lane, and never
// passes through LDS.  The sample order inside a wave is permuted so that each lane reads 4 CONSECUTIVE
// samples (32 B) of its column per step: a 16-lane group covers one full 128-byte line per variant.
// Waves are fully independent (each owns a contiguous sample range and all output tiles), so the kernel
// has no barriers; partial tiles go to a workspace and are summed in a fixed order by gene_stats_kernel.
// Only tiles with col-tile >= row-tile are computed (G'DG is symmetric).
// =====================================================================================================
// v_min_f64 / v_max_f64 without the canonicalisation pre-op fmin()/fmax() would add (inputs are finite genotypes)
__device__ __forceinline__ double raw_min(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double raw_max(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int MT, int CT, bool WEIGHTED, bool GUARD>
__device__ __forceinline__ void suffstat_step(const gcdp_t (&colp)[CT], gcdp_t vptr, long long off,
                                              long long nvalid, d4_t (&acc)[MT][CT], double (&cs)[MT],
                                              double (&cmn)[MT], double (&cmx)[MT], double (&f)[CT][4],
                                              unsigned long long* mask_ge, unsigned long long* mask_le, int lane,
                                              unsigned mask_bytes) {
  // (loads for this step were issued by the caller into f)
  double a[MT][4];
  if (WEIGHTED) {
    const d2_t v0 = *(gcd2p_t)(vptr + off);
    const d2_t v1 = *(gcd2p_t)(vptr + off + 2);
    const double vv[4] = {v0[0], v0[1], v1[0], v1[1]};
#pragma unroll
    for (int c = 0; c < MT; ++c)
#pragma unroll
      for (int l = 0; l < 4; ++l) a[c][l] = f[c][l] * vv[l];
  } else {
#pragma unroll
    for (int c = 0; c < MT; ++c)
#pragma unroll
      for (int l = 0; l < 4; ++l) a[c][l] = f[c][l];
  }
  unsigned long long wge = 0, wle = 0;
#pragma unroll
  for (int c = 0; c < MT; ++c) {
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const double g = f[c][l];
      cs[c] += g;
      if (GUARD) {
        if ((long long)l < nvalid) {
          cmn[c] = raw_min(cmn[c], g);
          cmx[c] = raw_max(cmx[c], g);
        }
      } else {
        cmn[c] = raw_min(cmn[c], g);
        cmx[c] = raw_max(cmx[c], g);
      }
      const unsigned long long bge = __ballot(g >= 1.0);
      const unsigned long long ble = __ballot(g <= 1.0);
      if (lane == c * 4 + l) {
        wge = bge;
        wle = ble;
      }
    }
  }
  // Lanes 0 .. 4*MT-1 hold this step's ballots.  Store them through a buffer descriptor whose num_records covers
  // exactly those lanes: the hardware drops the out-of-range lanes, so there is no exec-masked branch in the loop
  // (a branch here makes the compiler wait vmcnt(0) every step and serialises the load ring).
  {
    typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
    // the descriptor must live in SGPRs: make the (wave-uniform) base addresses provably uniform, otherwise the
    // compiler wraps each store in a waterfall loop
    auto uniform_ptr = [](unsigned long long* p) {
      const unsigned long long a = (unsigned long long)p;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
      const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
      return (void*)(((unsigned long long)hi << 32) | lo);
    };
    // (mask_bytes = MT * 32, or 0 for a gene without mask planes — a hard-call gene handed back to this kernel, whose
    // burden sums come from burden_fallback_kernel: every store is then out of range and dropped)
    const __amdgpu_buffer_rsrc_t rge = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(mask_ge), 0, mask_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rle = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(mask_le), 0, mask_bytes, 0x00020000);
    u2_t dge, dle;
    dge[0] = (unsigned)wge;
    dge[1] = (unsigned)(wge >> 32);
    dle[0] = (unsigned)wle;
    dle[1] = (unsigned)(wle >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(dge, rge, lane * 8, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(dle, rle, lane * 8, 0, 0);
  }
#pragma unroll
  for (int l = 0; l < 4; ++l) {
#pragma unroll
    for (int r = 0; r < MT; ++r) {
#pragma unroll
      for (int c = r; c < CT; ++c) {
        acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r][l], f[c][l], acc[r][c], 0, 0, 0);
      }
    }
  }
  if (!GUARD) {
    // One wave per SIMD: the VALU work of this step (column statistics, ballots) only overlaps the matrix pipe
    // if it is issued BETWEEN the MFMAs (each occupies the pipe 64 cycles).  Ask the scheduler for
    // "1 MFMA, then a handful of VALU" groups instead of an MFMA cluster followed by a VALU cluster.
    constexpr int kMfma = 4 * (MT * CT - MT * (MT - 1) / 2);
    constexpr int kValuPer = (MT * 4 * 9 + kMfma - 1) / kMfma + 1;
#pragma unroll
    for (int i = 0; i < kMfma; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, kValuPer, 0);  // a few VALU
    }
  }
}

template <int CT>
__device__ __forceinline__ void suffstat_load(const gcdp_t (&colp)[CT], long long off, double (&f)[CT][4]) {
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const d2_t x0 = *(gcd2p_t)(colp[c] + off);
    const d2_t x1 = *(gcd2p_t)(colp[c] + off + 2);
    f[c][0] = x0[0];
    f[c][1] = x0[1];
    f[c][2] = x1[0];
    f[c][3] = x1[1];
  }
}

template <int MT, int CT, bool WEIGHTED, int DEPTH>
__device__ __forceinline__ void suffstat_body(const GeneDesc& gd, const NullDev& nd, long long N, long long ld, int d,
                                              int wpart) {
  const int lane = threadIdx.x & 63;
  if (wpart >= gd.n_wparts) return;
  const long long nsteps = ld >> 4;
  const long long s_begin = (long long)wpart * gd.steps_per_wpart;
  long long s_end = s_begin + gd.steps_per_wpart;
  if (s_end > nsteps) s_end = nsteps;
  const int M = gd.M;
  // per-lane column pointers
  gcdp_t colp[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int j = c * 16 + (lane & 15);
    const double* p;
    if (j < M)
      p = gd.G + (long long)j * ld;
    else if (j < M + d)
      p = nd.X + (long long)(j - M) * ld;
    else if (j == M + d)
      p = nd.rr;
    else
      p = nd.zeros;
    colp[c] = as_global(p);
  }
  const gcdp_t vglob = as_global(nd.v);
  d4_t acc[MT][CT];
#pragma unroll
  for (int r = 0; r < MT; ++r)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[r][c] = d4_t{0.0, 0.0, 0.0, 0.0};
  double cs[MT], cmn[MT], cmx[MT];
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    cs[c] = 0.0;
    cmn[c] = INFINITY;
    cmx[c] = -INFINITY;
  }
  const long long koff = (long long)(lane >> 4) * 4;
  unsigned long long* mge = gd.masks;
  unsigned long long* mle = gd.masks + nsteps * MT * 4;
  const unsigned mask_bytes = gd.masks ? (unsigned)(MT * 32) : 0u;
  // steps whose 16 samples are all < N need no guard
  const long long full_steps = N >> 4;
  // Register ring of DEPTH step buffers: while step s is multiplied, the loads of the next DEPTH-1 steps are in
  // flight (one wave per SIMD at these register counts, so memory-level parallelism has to come from the wave
  // itself: (DEPTH-1) x CT x 2 KiB outstanding per wave).  DEPTH = 3 for genes up to 64 variants, 2 above.
  // Steps whose 16 samples are all < N run unguarded in a branch-free loop (prefetch indices are clamped instead
  // of tested); the single possibly-partial last step of the gene is handled after the loop.
  const long long s_lim = (s_end < full_steps) ? s_end : full_steps;
  auto step = [&](double (&f)[CT][4], long long s) {
    suffstat_step<MT, CT, WEIGHTED, false>(colp, vglob, s * 16 + koff, 4, acc, cs, cmn, cmx, f, mge + s * MT * 4,
                                           mle + s * MT * 4, lane, mask_bytes);
  };
  auto load = [&](double (&f)[CT][4], long long s) {
    const long long sc = (s < s_lim) ? s : s_lim - 1;  // clamp: a redundant reload near the end, never out of range
    suffstat_load<CT>(colp, sc * 16 + koff, f);
  };
  long long s = s_begin;
  if (s < s_lim) {
    if constexpr (DEPTH == 3) {
      double f0[CT][4], f1[CT][4], f2[CT][4];
      load(f0, s);
      load(f1, s + 1);
      for (; s + 2 < s_lim; s += 3) {
        load(f2, s + 2);
        step(f0, s);
        load(f0, s + 3);
        step(f1, s + 1);
        load(f1, s + 4);
        step(f2, s + 2);
      }
      if (s < s_lim) step(f0, s);  // 0, 1 or 2 steps left; their data is already in f0 / f1
      if (s + 1 < s_lim) step(f1, s + 1);
    } else {
      double f0[CT][4], f1[CT][4];
      load(f0, s);
      for (; s + 1 < s_lim; s += 2) {
        load(f1, s + 1);
        step(f0, s);
        load(f0, s + 2);
        step(f1, s + 1);
      }
      if (s < s_lim) step(f0, s);
    }
  }
  if (s_end > full_steps && full_steps >= s_begin) {  // the gene's last, partially filled step
    double fg[CT][4];
    const long long sg = full_steps, off = sg * 16 + koff;
    suffstat_load<CT>(colp, off, fg);
    suffstat_step<MT, CT, WEIGHTED, true>(colp, vglob, off, N - off, acc, cs, cmn, cmx, fg, mge + sg * MT * 4,
                                          mle + sg * MT * 4, lane, mask_bytes);
  }
  // ---- write this wave's partial tiles: element (row, col) -> parts[row*Cp + col] --------------------
  double* out = gd.parts + (long long)wpart * gd.Mp * gd.Cp;
#pragma unroll
  for (int r = 0; r < MT; ++r) {
#pragma unroll
    for (int c = r; c < CT; ++c) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r * 16 + (lane >> 4) + 4 * i;
        const int col = c * 16 + (lane & 15);
        out[(long long)row * gd.Cp + col] = acc[r][c][i];
      }
    }
  }
  // ---- column sum / min / max: combine the 4 sample groups (lane>>4) --------------------------------
  double* cst = gd.colstat + (long long)wpart * 3 * gd.Mp;
#pragma unroll
  for (int c = 0; c < MT; ++c) {
    double s = cs[c], mn = cmn[c], mx = cmx[c];
    s += __shfl_xor(s, 16, 64);
    mn = fmin(mn, __shfl_xor(mn, 16, 64));
    mx = fmax(mx, __shfl_xor(mx, 16, 64));
    s += __shfl_xor(s, 32, 64);
    mn = fmin(mn, __shfl_xor(mn, 32, 64));
    mx = fmax(mx, __shfl_xor(mx, 32, 64));
    if (lane < 16) {
      cst[c * 16 + lane] = s;
      cst[gd.Mp + c * 16 + lane] = mn;
      cst[2 * gd.Mp + c * 16 + lane] = mx;
    }
  }
}


// One body per tile configuration (MT row tiles x CT column tiles); the configurations are grouped by register
// budget into three kernels so that a batch needs three launches, not one per configuration — every launch ends
// with a tail in which the chip drains, and with the genes of a launch sorted widest-first the tail of a big
// launch is short:
//   group 0  <= 128 registers, 4 waves / SIMD   (1,1) (1,2) (2,2)            M <= 28 (unweighted)
//   group 1  <= 256 registers, 2 waves / SIMD   (2,3) (3,3) (3,4) (4,4) (4,5) M <= 64   [+ weighted (2,2)]
//   group 2  <= 512 registers, 1 wave  / SIMD   (5,5) (5,6) (6,6) (6,7)       M <= 96
// Inside a kernel the gene's configuration selects the body (a wave-uniform switch); the register allocation of
// the kernel is the largest of its group.  Grid = (wave-parts, genes of the group), one wave per workgroup.
template <int MT, int CT, bool WEIGHTED>
__device__ __forceinline__ void suffstat_class(const GeneDesc& gd, const NullDev& nd, long long N, long long ld,
                                               int d, int wpart) {
  // ring depth: 3 wherever the registers allow it without dropping an occupancy step (see tools/kernel_regs.sh)
  constexpr int kDepth = (CT <= 3)         ? 3
                         : (MT * CT <= 16) ? (WEIGHTED ? 2 : 3)
                         : (MT * CT <= 20) ? 2   // (4,5): fits 256 registers, two waves per SIMD
                         : (CT <= 5)       ? 3
                                           : 2;
  suffstat_body<MT, CT, WEIGHTED, kDepth>(gd, nd, N, ld, d, wpart);
}

__host__ __device__ constexpr int suffstat_group(int MT, int CT, bool weighted) {
  return (MT * CT > 20) ? 2 : ((MT * CT <= 4 && !(weighted && MT == 2)) ? 0 : 1);
}

// Two ways to launch it:
//   list == nullptr   grid (wave-parts, genes): workgroup (x, y) computes wave-part x of genes[y];
//   list != nullptr   a WORK LIST written on the device (gene_flags_hc_kernel): list[0] = number of genes the hard-call
//                     kernel handed back, list[4 ..] = their indices into `genes`; any grid — the workgroups loop over
//                     the (gene, wave-part) items, and leave at once when the list is empty (the usual case).
template <int GROUP, bool WEIGHTED>
__global__ __launch_bounds__(64, GROUP == 0 ? 4 : (GROUP == 1 ? 2 : 1)) void gene_suffstat_mfma(
    const GeneDesc* __restrict__ genes, const int* __restrict__ list, int n_wparts, NullDev nd, long long N,
    long long ld, int d) {
  long long item = (long long)blockIdx.x + (long long)blockIdx.y * gridDim.x;
  const long long stride = (long long)gridDim.x * gridDim.y;
  const long long n_items = list ? (long long)list[0] * n_wparts : 0;
  do {
    int gi = blockIdx.y, wpart = blockIdx.x;
    if (list) {
      if (item >= n_items) return;
      gi = list[4 + (int)(item / n_wparts)];
      wpart = (int)(item % n_wparts);
    }
    const GeneDesc gd = genes[gi];
    const int cls = gd.MT * 8 + gd.CT;
#define RVT_CLASS(mt, ct)                                                \
  case mt * 8 + ct:                                                      \
    if constexpr (suffstat_group(mt, ct, WEIGHTED) == GROUP)             \
      suffstat_class<mt, ct, WEIGHTED>(gd, nd, N, ld, d, wpart);         \
    break
    switch (cls) {
      RVT_CLASS(1, 1);
      RVT_CLASS(1, 2);
      RVT_CLASS(2, 2);
      RVT_CLASS(2, 3);
      RVT_CLASS(3, 3);
      RVT_CLASS(3, 4);
      RVT_CLASS(4, 4);
      RVT_CLASS(4, 5);
      RVT_CLASS(5, 5);
      RVT_CLASS(5, 6);
      RVT_CLASS(6, 6);
      RVT_CLASS(6, 7);
      default:
        break;
    }
#undef RVT_CLASS
    item += stride;
  } while (list);
}

// ---- genes wider than 6 row tiles (M > 96): the tile grid is cut into panels of up to 4 x 4 tiles -----------
// blockIdx.z enumerates the panels (pr, pc), pc >= pr.  A diagonal panel (pr == pc) holds the same 64 columns on
// both sides (one set of loads) and also produces the column statistics and bit masks of those columns; an
// off-diagonal panel loads its 64 "row" columns and its 64 "column" columns separately.  G is therefore read
// about (number of panel rows) times — wide genes are compute-bound anyway (intensity grows with M).
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void gene_suffstat_panel(const GeneDesc* __restrict__ genes, NullDev nd, long long N,
                                                           long long ld, int d) {
  const GeneDesc gd = genes[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int wpart = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (wpart >= gd.n_wparts) return;
  const int nPR = (gd.MT + 3) / 4, nPC = (gd.CT + 3) / 4;
  int pr = 0, pc = 0, z = blockIdx.z;
  for (pr = 0; pr < nPR; ++pr) {
    const int cnt = nPC - pr;
    if (z < cnt) {
      pc = pr + z;
      break;
    }
    z -= cnt;
  }
  if (pr >= nPR) return;
  const bool diag = (pr == pc);
  const int r0 = pr * 4, c0 = pc * 4, M = gd.M, MT = gd.MT;
  const long long nsteps = ld >> 4;
  const long long s_begin = (long long)wpart * gd.steps_per_wpart;
  long long s_end = s_begin + gd.steps_per_wpart;
  if (s_end > nsteps) s_end = nsteps;
  auto column = [&](int j) -> const double* {
    if (j < M) return gd.G + (long long)j * ld;
    if (j < M + d) return nd.X + (long long)(j - M) * ld;
    if (j == M + d) return nd.rr;
    return nd.zeros;
  };
  gcdp_t pa[4], pb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ja = (r0 + t) * 16 + (lane & 15);
    pa[t] = as_global((r0 + t < MT && ja < M) ? gd.G + (long long)ja * ld : nd.zeros);
    pb[t] = as_global((c0 + t < gd.CT) ? column((c0 + t) * 16 + (lane & 15)) : nd.zeros);
  }
  const gcdp_t vglob = as_global(nd.v);
  d4_t acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = d4_t{0.0, 0.0, 0.0, 0.0};
  double cs[4], cmn[4], cmx[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    cs[t] = 0.0;
    cmn[t] = INFINITY;
    cmx[t] = -INFINITY;
  }
  const long long koff = (long long)(lane >> 4) * 4;
  unsigned long long* mge = gd.masks;
  unsigned long long* mle = gd.masks + nsteps * MT * 4;
  const unsigned mask_bytes = gd.masks ? (unsigned)(MT * 32) : 0u;
  const int nrow = (MT - r0 < 4) ? MT - r0 : 4;  // row tiles of this panel that exist
  for (long long s = s_begin; s < s_end; ++s) {
    const long long off = s * 16 + koff;
    double fa[4][4], fb[4][4], a[4][4];
    suffstat_load<4>(pa, off, fa);
    // B side: its own loads, except in a diagonal panel where a tile made only of genotype columns is the very
    // register set already loaded for the A side (a tile that also holds X / rr columns differs: A rows are G only)
    if (!diag || (c0 + 4) * 16 > M) suffstat_load<4>(pb, off, fb);
    double vv[4] = {1.0, 1.0, 1.0, 1.0};
    if (WEIGHTED) {
      const d2_t v0 = *(gcd2p_t)(vglob + off);
      const d2_t v1 = *(gcd2p_t)(vglob + off + 2);
      vv[0] = v0[0];
      vv[1] = v0[1];
      vv[2] = v1[0];
      vv[3] = v1[1];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        a[t][l] = WEIGHTED ? fa[t][l] * vv[l] : fa[t][l];
        if (diag && (c0 + 4) * 16 <= M) fb[t][l] = fa[t][l];
      }
    if (diag) {
      unsigned long long wge = 0, wle = 0;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          const double g = fa[t][l];
          cs[t] += g;
          if (off + l < N) {
            cmn[t] = raw_min(cmn[t], g);
            cmx[t] = raw_max(cmx[t], g);
          }
          const unsigned long long bge = __ballot(g >= 1.0);
          const unsigned long long ble = __ballot(g <= 1.0);
          if (lane == t * 4 + l) {
            wge = bge;
            wle = ble;
          }
        }
      if (lane < nrow * 4) {
        mge[s * MT * 4 + r0 * 4 + lane] = wge;
        mle[s * MT * 4 + r0 * 4 + lane] = wle;
      }
    }
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (diag && c < r) continue;
          acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r][l], fb[c][l], acc[r][c], 0, 0, 0);
        }
  }
  double* out = gd.parts + (long long)wpart * gd.Mp * gd.Cp;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (diag && c < r) continue;
      if (r0 + r < MT && c0 + c < gd.CT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (r0 + r) * 16 + (lane >> 4) + 4 * i;
          const int col = (c0 + c) * 16 + (lane & 15);
          out[(long long)row * gd.Cp + col] = acc[r][c][i];
        }
      }
    }
  if (diag) {
    double* cst = gd.colstat + (long long)wpart * 3 * gd.Mp;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      double sm = cs[t], mn = cmn[t], mx = cmx[t];
      sm += __shfl_xor(sm, 16, 64);
      mn = fmin(mn, __shfl_xor(mn, 16, 64));
      mx = fmax(mx, __shfl_xor(mx, 16, 64));
      sm += __shfl_xor(sm, 32, 64);
      mn = fmin(mn, __shfl_xor(mn, 32, 64));
      mx = fmax(mx, __shfl_xor(mx, 32, 64));
      if (lane < 16 && r0 + t < MT) {
        cst[(r0 + t) * 16 + lane] = sm;
        cst[gd.Mp + (r0 + t) * 16 + lane] = mn;
        cst[2 * gd.Mp + (r0 + t) * 16 + lane] = mx;
      }
    }
  }
}


}  // namespace rvt
