// rvtests_amd — HIP kernels for gfx950 (MI355X).  Included once by rvt_engine.hip.
//
//  gene_suffstat_mfma<MT,CT,W>  fp64-MFMA sufficient statistics  R = G'·D·[G | X | rr]  + exact column
//                               sum/min/max + per-sample ">=1" / "<=1" bit masks, ONE pass over G
//  gene_flags_kernel            flip / polymorphic flags per 16-variant block
//  burden_collapse_kernel       cmcCollapse / zegginiCollapse from the bit masks + score partial sums
//  gene_stats_kernel            flip algebra, weights, eigenvalues, SKAT-O moments (one workgroup / gene)
//  gene_pvalue_kernel           Davies / Liu / QAGS (one wave / gene, one lane / quadrature abscissa)
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include "rvt_pvalue.h"
#include "rvt_mvn.h"

#include "suffstat_kernels.hip.h"
#include "suffstat_hc.hip.h"
#include "suffstat_hcw.hip.h"

namespace rvt {

// =====================================================================================================
// flip / polymorphic flags per 16-variant block (needed by the burden kernel before gene_stats runs)
//   flip:  column sum > N       convertToMinorAlleleCount   src/DataConsolidator.cpp:46-69
//   poly:  min != max           isMonomorphicMarker          src/DataConsolidator.cpp:94-116
// =====================================================================================================
#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(64) void gene_flags_kernel(const GeneDesc* __restrict__ genes, long long N) {
  const GeneDesc gd = genes[blockIdx.x];
  // (a hard-call gene: only when it was handed back and gene_suffstat_mfma has computed it — three-row statistics)
  if (gd.hc && gd.flags[2 * gd.MT + 1] == 0) return;
  const int tid = threadIdx.x;  // blockDim.x == 64: one wave
  for (int base = 0; base < gd.Mp; base += 64) {
    const int j = base + tid;
    double s = 0.0, mn = INFINITY, mx = -INFINITY;
    if (j < gd.M) {
      for (int p = 0; p < gd.n_wparts; ++p) {
        const double* c = gd.colstat + (long long)p * 3 * gd.Mp;
        s += c[j];
        mn = fmin(mn, c[gd.Mp + j]);
        mx = fmax(mx, c[2 * gd.Mp + j]);
      }
    }
    const bool flip = (j < gd.M) && !(s <= (double)N);
    const bool poly = (j < gd.M) && !(mn == mx);
    const unsigned long long bf = __ballot(flip), bp = __ballot(poly);
    if (tid < 4) {
      const int b = (base >> 4) + tid;  // 16-variant block
      if (b < gd.MT) {
        gd.flags[b] = (unsigned short)((bf >> (16 * tid)) & 0xffffu);
        gd.flags[gd.MT + b] = (unsigned short)((bp >> (16 * tid)) & 0xffffu);
      }
    }
  }
}
#endif  // RVT_K_ENGINE

// =====================================================================================================
// K1b: collapsed burden genotypes and their score-test partial sums.
//   cmcCollapse / zegginiCollapse (src/Model.cpp:73-89,115-130): a variant "counts" for a sample when
//   (int)g' > 0 on the flipped genotype g': g >= 1 for an unflipped column, g <= 1 for a flipped one.
//   Per sample:  n = popcount(((ge & ~flip) | (le & flip)) & poly);  c_cmc = (n > 0), c_zeg = n.
//   Partial sums per block and test: U = Σ c·res, Σ w c², #(c != 0), Σ w c x_k   (w = v if binary else 1)
// One thread = one (16-sample step, l) pair, i.e. the 4 samples whose bits share a ballot word, looped over the
// genes of the launch so X/res/v are read once per sample; each mask word is read by exactly one thread and a
// wave consumes a contiguous 16-step run of the mask array.  The 2 x (3+d) partial sums of a wave are reduced by
// a halving butterfly (the value set is split between the two halves of the exchange at every stage), which
// needs ~(3+d)*2 + 6 cross-lane moves instead of 6 per value.
// =====================================================================================================
constexpr int kBurdenSPB = 1024;  // samples per block (256 threads x 4)

template <int DMAX>
__global__ __launch_bounds__(256) void burden_collapse_kernel(const GeneDesc* __restrict__ genes, int n_genes,
                                                              int genes_per_group, NullDev nd, long long N,
                                                              long long ld, int d, int binary, unsigned tests) {
  constexpr int NV = 2 * (3 + DMAX);                       // values reduced per gene
  constexpr int NP = NV <= 16 ? 16 : (NV <= 32 ? 32 : 64); // padded to a power of two for the butterfly
  __shared__ double red[4][NV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long base = (long long)blockIdx.x * kBurdenSPB;
  const long long nsteps = ld >> 4;
  const int rl = 3 + d;
  const long long step = (base >> 4) + (tid >> 2);
  const int l = tid & 3;
  const bool step_ok = step < nsteps;
  double xr[4][DMAX], rres[4], wv[4];
  bool valid[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // q = the 4-sample group (kk) inside the step
    const long long smp = step * 16 + q * 4 + l;
    valid[q] = smp < N;
    const long long sidx = valid[q] ? smp : 0;
    rres[q] = valid[q] ? nd.res[sidx] : 0.0;
    wv[q] = valid[q] ? (binary ? nd.v[sidx] : 1.0) : 0.0;
#pragma unroll
    for (int k = 0; k < DMAX; ++k) xr[q][k] = (valid[q] && k < d) ? nd.X[(long long)k * ld + sidx] : 0.0;
  }
  // blockIdx.y = gene group: groups of `genes_per_group` genes run as concurrent workgroups of ONE launch (more waves
  // in flight than one launch per group; the kernel is latency-bound)
  const int g_begin = blockIdx.y * genes_per_group;
  const int g_end = (g_begin + genes_per_group < n_genes) ? g_begin + genes_per_group : n_genes;
  for (int g = g_begin; g < g_end; ++g) {
    const GeneDesc gd = genes[g];
    const int MT = gd.MT;
    const unsigned long long* mge = gd.masks + (step * MT) * 4 + l;
    const unsigned long long* mle = mge + nsteps * MT * 4;
    int n[4] = {0, 0, 0, 0};
    if (step_ok) {
      for (int c = 0; c < MT; ++c) {
        const unsigned long long ge = mge[c * 4], le = mle[c * 4];
        unsigned long long fl = gd.flags[c], po = gd.flags[MT + c];
        fl |= fl << 16;
        fl |= fl << 32;
        po |= po << 16;
        po |= po << 32;
        const unsigned long long hit = ((ge & ~fl) | (le & fl)) & po;
        n[0] += __popc((unsigned)hit & 0xffffu);
        n[1] += __popc((unsigned)hit >> 16);
        n[2] += __popc((unsigned)(hit >> 32) & 0xffffu);
        n[3] += __popc((unsigned)(hit >> 48));
      }
    }
    double val[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) val[k] = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nq = valid[q] ? n[q] : 0;
      const double cv[2] = {nq > 0 ? 1.0 : 0.0, (double)nq};
      if (valid[q]) {
        const long long smp = step * 16 + q * 4 + l;
        if (gd.dbg_cmc) gd.dbg_cmc[smp] = cv[0];
        if (gd.dbg_zeg) gd.dbg_zeg[smp] = cv[1];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const double c = cv[t];
        val[t * (3 + DMAX) + 0] += c * rres[q];
        val[t * (3 + DMAX) + 1] += (c * wv[q]) * c;
        val[t * (3 + DMAX) + 2] += (c != 0.0) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < DMAX; ++k) val[t * (3 + DMAX) + 3 + k] += (c * wv[q]) * xr[q][k];
      }
    }
    // halving butterfly in a fixed order: after the stage with lane distance D, a lane holds the sums over its
    // D-pair of the half of the value set selected by its bit; when one value is left the remaining stages are
    // plain exchanges.  Lane L ends with the wave total of value index bitreverse-free id computed below.
    int width = NP;
#pragma unroll
    for (int D = 32; D >= 1; D >>= 1) {
      const bool upper = (lane & D) != 0;
      if (width > 1) {
        const int h = width >> 1;
#pragma unroll
        for (int k = 0; k < NP / 2; ++k) {
          if (k < h) {
            const double keep = upper ? val[k + h] : val[k];
            const double send = upper ? val[k] : val[k + h];
            val[k] = keep + __shfl_xor(send, D, 64);
          }
        }
        width = h;
      } else {
        val[0] += __shfl_xor(val[0], D, 64);
      }
    }
    // value index held by this lane: stage i (D = 32 >> i) chose the upper half of a width NP >> i set
    {
      int idx = 0, w = NP, D = 32;
      while (w > 1) {
        w >>= 1;
        if (lane & D) idx += w;
        D >>= 1;
      }
      // lanes that differ only in bits below the last halving stage hold the same value; one of them writes
      const int low_mask = (NP >= 64) ? 0 : ((64 / NP) - 1);
      if ((lane & low_mask) == 0) {
        const int t = idx / (3 + DMAX), k = idx % (3 + DMAX);
        if (idx < NV && k < rl) red[wave][t * rl + k] = val[0];
      }
    }
    __syncthreads();
    if (tid < 2 * rl) {
      const double s = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
      gd.bparts[(long long)blockIdx.x * 2 * rl + tid] = s;
    }
    __syncthreads();
  }
}

// =====================================================================================================
// Hard-call path (suffstat_hc.hip.h): flags + verification of what the in-pass burden collapse assumed.
//   flags[0 .. MT)      flip bits per 16-variant block (column sum > N)
//   flags[MT .. 2 MT)   polymorphic bits
//   flags[2 MT]         1 = the burden sums of gene_suffstat_hc are NOT valid: a column's flip was predicted wrongly
//                       (allele frequency vs exact column sum), a monomorphic column was counted by the collapse
//                       (all-1 column; all-0 and all-2 columns never count), or a column's imputed value counts
//                       ((int)g' > 0) where the in-pass collapse skips masked entries -> burden_fallback_kernel
//   flags[2 MT + 1]     1 = the block is not "hard calls + one imputed value per column" (dosages, -inf): the statistics
//                       of gene_suffstat_hc are void.  The gene is HANDED BACK: the conditional launch of
//                       gene_suffstat_mfma that follows on the same stream computes it (every other workgroup of that
//                       launch leaves at once), gene_flags_kernel derives its flags from the three-row statistics, its
//                       burden sums come from burden_fallback_kernel, and gene_assemble reads it as a general-path gene
// =====================================================================================================
// lists: [0] number of handed-back genes, [1] number of genes whose burden sums are redone, [4 ..) / [4 + n_genes ..) their
// indices (the work lists of the conditional general-kernel launch and of burden_fallback_kernel; zeroed by the host).
#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(64) void gene_flags_hc_kernel(const GeneDesc* __restrict__ genes, long long N,
                                                           int* __restrict__ lists, int n_genes) {
  const GeneDesc gd = genes[blockIdx.x];
  const int tid = threadIdx.x;
  bool bad = false, rerun = false;
  for (int p = tid; p < gd.n_wparts; p += 64) rerun |= gd.wflags && (gd.wflags[p] & 2u);  // an entry with code 3 (-inf)
  for (int base = 0; base < gd.Mp; base += 64) {
    const int j = base + tid;
    double s = 0.0, mn = INFINITY, mx = -INFINITY, cm = 0.0;
    unsigned long long orb = 0ull, andb = ~0ull;
    if (j < gd.M) {
      for (int p = 0; p < gd.n_wparts; ++p) {
        const double* c = gd.colstat + (long long)p * kHcColstatRows * gd.Mp;
        s += c[j];
        mn = fmin(mn, c[gd.Mp + j]);
        mx = fmax(mx, c[2 * gd.Mp + j]);
        cm += c[3 * gd.Mp + j];
        orb |= reinterpret_cast<const unsigned long long*>(c)[4 * gd.Mp + j];
        andb &= reinterpret_cast<const unsigned long long*>(c)[5 * gd.Mp + j];
      }
    }
    const bool in = j < gd.M;
    // masked entries: one bit pattern per column (OR == AND), a finite value in [0, 2] — the mean imputeGenotypeToMean
    // wrote (src/DataConsolidator.cpp:217-245).  Anything else (dosages) is the general kernel's.
    const double mu = rvt_bits_to_double(orb);
    const bool masked = in && cm > 0.0;
    if (masked) {
      if (orb != andb || !(mu >= 0.0 && mu <= 2.0)) rerun = true;
      s += cm * mu;
      mn = fmin(mn, mu);
      mx = fmax(mx, mu);
    }
    // (lattice dosages, hc == 2: s is the integer sum of K = den x the column sum — compared exactly)
    const bool flip = in && !(s <= ((gd.hc == 2 || gd.hc == 4) ? (double)N * gd.lat_den : (double)N));
    const bool poly = in && !(mn == mx);
    const bool pred = in && (j >> 4) < 8 && ((gd.pflip[(j >> 4) & 7] >> (j & 15)) & 1);
    const bool counted_mono = in && !poly && (pred ? mn != 2.0 : mn != 0.0);
    // the in-pass collapse never counts a masked entry: wrong when (int)g' > 0 for the imputed value
    const bool masked_counts = masked && poly && (flip ? mu <= 1.0 : mu >= 1.0);
    // (hc == 3, packed rows: the kernel had the exact flips, polymorphic flags and imputed values — nothing to verify)
    if (gd.hc != 3) bad |= (flip != pred) || counted_mono || masked_counts;
    const unsigned long long bf = __ballot(flip), bp = __ballot(poly);
    if (tid < 4) {
      const int b = (base >> 4) + tid;
      if (b < gd.MT) {
        gd.flags[b] = (unsigned short)((bf >> (16 * tid)) & 0xffffu);
        gd.flags[gd.MT + b] = (unsigned short)((bp >> (16 * tid)) & 0xffffu);
      }
    }
  }
  const bool any = __any(bad), anyr = __any(rerun);
  if (tid == 0) {
    gd.flags[2 * gd.MT] = (any || anyr) ? 1 : 0;
    gd.flags[2 * gd.MT + 1] = anyr ? 1 : 0;
    if (anyr) lists[4 + atomicAdd(&lists[0], 1)] = blockIdx.x;
    if (any || anyr) lists[4 + n_genes + atomicAdd(&lists[1], 1)] = blockIdx.x;
  }
}
#endif  // RVT_K_ENGINE

// Burden partial sums of a hard-call gene straight from its genotype block with the ACTUAL flags (rare: see above).
// The genes come from the device work list gene_flags_hc_kernel wrote (lists[1] entries at lists[4 + n_genes ..)); a
// fixed grid of kFallbackGrid workgroups loops over the (gene, wave-part) items and leaves at once when the list is
// empty.  Writes the same records gene_suffstat_hc writes: bparts[part][test][..].
constexpr int kFallbackGrid = 1024;
template <int DMAX>
__global__ __launch_bounds__(256) void burden_fallback_kernel(const GeneDesc* __restrict__ genes,
                                                              const int* __restrict__ lists, int n_genes, int n_wparts,
                                                              NullDev nd, long long N, long long ld, int d, int binary) {
  const long long n_items = (long long)lists[1] * n_wparts;
  for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
  const GeneDesc gd = genes[lists[4 + n_genes + (int)(item / n_wparts)]];
  if (!gd.bparts) continue;  // (uniform over the workgroup)
  constexpr int NV = 2 * (3 + DMAX);
  __shared__ double red[4][NV];
  __shared__ int kcol[RVT_MAX_VARIANTS];  // kept (polymorphic) columns: index | flip << 30
  __shared__ int nkept;
  const int tid = threadIdx.x;
  {
  const int part = (int)(item % n_wparts);
  if (tid == 0) {  // (one pass over <= 96 flag bits; the sample loop below then touches no flag)
    int m = 0;
    for (int j = 0; j < gd.M; ++j) {
      const int b = j >> 4, bit = j & 15;
      if ((gd.flags[gd.MT + b] >> bit) & 1) kcol[m++] = j | (((gd.flags[b] >> bit) & 1) << 30);
    }
    nkept = m;
  }
  __syncthreads();
  const int m_kept = nkept;
  const long long s0 = (long long)part * gd.steps_per_wpart * 16;
  long long s1 = s0 + (long long)gd.steps_per_wpart * 16;
  if (s1 > N) s1 = N;
  double val[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) val[k] = 0.0;
  for (long long i = s0 + tid; i < s1; i += 256) {
    int n = 0;
    int a = 0;
    for (; a + 4 <= m_kept; a += 4) {  // four independent loads in flight
      const int c0 = kcol[a], c1 = kcol[a + 1], c2 = kcol[a + 2], c3 = kcol[a + 3];
      const double g0 = gd.G[(long long)(c0 & 0x3fffffff) * ld + i], g1 = gd.G[(long long)(c1 & 0x3fffffff) * ld + i],
                   g2 = gd.G[(long long)(c2 & 0x3fffffff) * ld + i], g3 = gd.G[(long long)(c3 & 0x3fffffff) * ld + i];
      n += ((int)((c0 >> 30) ? 2.0 - g0 : g0) > 0) + ((int)((c1 >> 30) ? 2.0 - g1 : g1) > 0) +
           ((int)((c2 >> 30) ? 2.0 - g2 : g2) > 0) + ((int)((c3 >> 30) ? 2.0 - g3 : g3) > 0);
    }
    for (; a < m_kept; ++a) {
      const int c0 = kcol[a];
      const double g0 = gd.G[(long long)(c0 & 0x3fffffff) * ld + i];
      n += ((int)((c0 >> 30) ? 2.0 - g0 : g0) > 0);
    }
    const double cv[2] = {n > 0 ? 1.0 : 0.0, (double)n};
    const double r = nd.res[i], w = binary ? nd.v[i] : 1.0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const double c = cv[t];
      val[t * (3 + DMAX) + 0] += c * r;
      val[t * (3 + DMAX) + 1] += (c * w) * c;
      val[t * (3 + DMAX) + 2] += (c != 0.0) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < DMAX; ++k)
        if (k < d) val[t * (3 + DMAX) + 3 + k] += (c * w) * nd.X[(long long)k * ld + i];
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {  // fixed-order butterfly inside the wave, then the four waves in order
    double v = val[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) red[tid >> 6][k] = v;
  }
  __syncthreads();
  const int rl = 3 + d;
  if (tid < 2 * rl) {
    const int t = tid / rl, k = tid % rl, idx = t * (3 + DMAX) + k;
    gd.bparts[(long long)part * 2 * rl + tid] = ((red[0][idx] + red[1][idx]) + red[2][idx]) + red[3][idx];
  }
  __syncthreads();
  }
  }
}

// =====================================================================================================
// K3a: per-gene assembly (reduce partials, flags, flip algebra, projection, weights, Q, tau, burden
//      statistics), one 256-thread workgroup per gene.
// K3b: one workgroup per (dense reduction, gene): build the matrix in LDS, Householder tridiagonalisation
//      (one shared reduction serves the 12 SKAT-O eigenproblems, see rvt_gene.h stage B).
// K3c: one workgroup per (eigenproblem, gene): Sturm bisection of its tridiagonal, eigenvalue filter, moments.
// =====================================================================================================
// Round 6: step 1 of the assembly — R = the sum of the gene's wave-part images, 2 MB read per gene of M = 50 — on a grid of its
// own, (pieces of 1 024 entries, genes): the one workgroup per gene of gene_assemble_kernel issued at most 32 loads per
// thread and waited out the round trips (0.42 of its 0.72 ms per gene; 1.39 G of the kernel's 1.54 G wave cycles waiting,
// profiles/r5_pvprof.txt), eight times as many workgroups keep eight times as many loads in flight.  Every entry is still
// the sum over p = 0 .. P - 1 in that order: bit-identical to the loop it replaces (rvt_gene.h gene_assemble step 1).
constexpr int kReducePiece = 1024;
#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(256) void gene_reduce_parts_kernel(const GeneDesc* __restrict__ genes) {
  const GeneDesc gd = genes[blockIdx.y];
  const int Mp = gd.Mp, Cp = gd.Cp, total = Mp * Cp;
  const int idx = blockIdx.x * kReducePiece + (int)threadIdx.x;
  if (blockIdx.x * kReducePiece >= total) return;
  GeneScratch ws = gene_scratch_carve(gd.scratch, Mp, Cp);
  const bool handed_back = gd.hc && gd.flags[2 * gd.MT + 1];
  const double lat_den = ((gd.hc == 2 || gd.hc == 4) && !handed_back) ? gd.lat_den : 0.0;
  const double lat2 = lat_den > 0.0 ? lat_den * lat_den : 0.0;
  const size_t stride = (size_t)Mp * Cp;
  const int P = gd.n_wparts;
  double s4[4];
  bool use[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = idx + u * 256;
    s4[u] = 0.0;
    use[u] = i < total && ((i % Cp) >> 4) >= ((i / Cp) >> 4);
  }
  for (int p0 = 0; p0 < P; p0 += 8) {  // eight wave-parts x four entries: 32 loads issued before the first sum
    double t[8][4];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double* src = gd.parts + (size_t)(p0 + q) * stride;
#pragma unroll
      for (int u = 0; u < 4; ++u) t[q][u] = (use[u] && p0 + q < P) ? src[idx + u * 256] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p0 + q < P) s4[u] += t[q][u];  // (p ascending, as always)
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = idx + u * 256;
    if (i >= total) continue;
    double sv = s4[u];
    if (use[u] && lat2 > 0.0 && (i % Cp) < gd.M) sv /= lat2;  // lattice dosages: G'G = K'K / den^2 with one rounding
    ws.R[i] = sv;
  }
}
#endif  // RVT_K_ENGINE

#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(1024) void gene_assemble_kernel(const GeneDesc* __restrict__ genes,
                                                            const NullConsts* __restrict__ ncp, rvt_params prm,
                                                            unsigned tests, int n_bparts, const double* xscale,
                                                            int parts_reduced = 0) {
  __shared__ double red[64];
  __shared__ NullConsts nc;
  const GeneDesc gd = genes[blockIdx.x];
  if (threadIdx.x == 0) nc = *ncp;
  __syncthreads();
  Coop co{(int)threadIdx.x, (int)blockDim.x, red};
  GeneScratch ws = gene_scratch_carve(gd.scratch, gd.Mp, gd.Cp);
  const HcMasked hcm{gd.pq, gd.wflags, hc_pq_words(gd.MT), (gd.hc == 2 || gd.hc == 4) ? gd.lat_den : 0.0, gd.pqw, xscale};
  // a hard-call gene that was handed back holds the general kernel's statistics (three rows per wave-part, G'DG itself)
  const bool handed_back = gd.hc && gd.flags[2 * gd.MT + 1];
  const bool masks = gd.hc != 0 && !handed_back;  // (pq is null for the weighted and the lattice kernel: no masked tiles)
  gene_assemble(co, nc, gd.M, gd.Mp, gd.Cp, gd.parts, gd.n_wparts, gd.colstat,
                (tests & (RVT_TEST_CMC | RVT_TEST_ZEGGINI)) ? gd.bparts : nullptr, gd.n_bparts > 0 ? gd.n_bparts : n_bparts,
                gd.af, prm, tests, ws,
                gd.stats, gd.dbg_flip, gd.dbg_kept, masks ? &hcm : nullptr, handed_back ? kStatusHandedBack : 0u, parts_reduced != 0);
}
#endif  // RVT_K_ENGINE

#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(256) void gene_tridiag_kernel(const GeneDesc* __restrict__ genes,
                                                           const NullConsts* __restrict__ ncp, unsigned tests,
                                                           int lds_doubles) {
  extern __shared__ __attribute__((aligned(16))) double esm[];
  __shared__ double red[64];
  const GeneDesc gd = genes[blockIdx.y];
  const int which = blockIdx.x;
  Coop co{(int)threadIdx.x, (int)blockDim.x, red};
  GeneScratch ws = gene_scratch_carve(gd.scratch, gd.Mp, gd.Cp);
  const int m = gd.stats->n_poly;
  double* vec = esm;  // 4 * Mp doubles
  double* Bm = (4 * gd.Mp + m * m <= lds_doubles) ? esm + 4 * gd.Mp : ws.eig + (size_t)which * gd.Mp * gd.Mp;
  gene_tridiag(co, *ncp, which, gd.M, gd.Mp, tests, ws, Bm, vec, gd.stats);
}
#endif  // RVT_K_ENGINE

#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(128) void gene_spectrum_kernel(const GeneDesc* __restrict__ genes,
                                                            const NullConsts* __restrict__ ncp, unsigned tests) {
  extern __shared__ __attribute__((aligned(16))) double esm[];  // 4 * Mp doubles
  __shared__ double red[64];
  const GeneDesc gd = genes[blockIdx.y];
  Coop co{(int)threadIdx.x, (int)blockDim.x, red};
  GeneScratch ws = gene_scratch_carve(gd.scratch, gd.Mp, gd.Cp);
  gene_spectrum(co, *ncp, blockIdx.x, gd.M, gd.Mp, tests, ws, esm, gd.stats, gd.lambda);
}
#endif  // RVT_K_ENGINE
// round 5: ONE workgroup per gene walks the (problem, eigenvalue) tasks of all 13 problems (rvt_gene.h gene_spectrum_all);
// dynamic LDS: 39 * Mp + 64 doubles
#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(128) void gene_spectrum_all_kernel(const GeneDesc* __restrict__ genes,
                                                                const NullConsts* __restrict__ ncp, unsigned tests) {
  extern __shared__ __attribute__((aligned(16))) double esm[];
  __shared__ double red[64];
  __shared__ SpectrumMeta meta[kNEigen];
  const GeneDesc gd = genes[blockIdx.x];
  Coop co{(int)threadIdx.x, (int)blockDim.x, red};
  GeneScratch ws = gene_scratch_carve(gd.scratch, gd.Mp, gd.Cp);
  gene_spectrum_all(co, *ncp, gd.M, gd.Mp, tests, ws, esm, meta, gd.stats, gd.lambda);
}
#endif  // RVT_K_ENGINE

// =====================================================================================================
// MetaCov (src/Model.cpp:844-1004): the score covariances of a block of V consecutive variants are a by-product
// of the SAME sufficient statistics R = G'D[G | X | rr] the gene tests use — the block is run through the
// suffstat kernels as one "gene" and the two kernels below finish the algebra per (head, marker) pair:
//   quantitative (MetaCovUnrelatedQtl, :506-593): genotypes are centred, so
//       covXX(h,j) = (S_hj - s_h s_j / N) / sigma2,   covXZ_h = (T_h - (s_h/N) 1'Z) / sigma2
//   binary (MetaCovUnrelatedBinary, :694-778):  covXX = G'WG = S,  covXZ = G'WZ = T   (D = diag(v) already)
//   value(h,j) = covXX(h,j) - covXZ_h' covZZInv covXZ_j          (computeScaledXX, src/Model.h:3997-4005)
// s = exact column sums, T = G'DX from the X columns of R; covZZInv comes from the null model (engine side).
// =====================================================================================================
struct CovConsts {
  double zzinv[RVT_MAX_COV * RVT_MAX_COV];  // covZZInv, row-major d x d
  double zsum[RVT_MAX_COV];                 // 1'Z (quantitative centring)
  double inv_sigma2;                        // 1/sigma2 (quantitative), 1 (binary)
  double inv_n;                             // 1/N
  double c11;                               // family mode: u1' D u1
  double k1r;                               // family mode: u1' D uResid
  double af_denom;                          // family mode: u1' |lambda|^-1 u1 (FastLMM::GetAF)
  int d, binary;
  int fam;  // family mode (MetaCovFamQtl): R comes from the ROTATED block with D = 1/((|S|+delta) sigma2) and null
            // columns [U'X | u1]; genotypes are centred BEFORE the rotation, i.e. g~ - mean(g) u1:
            //   covXX = S_hj - m_h t1_j - m_j t1_h + m_h m_j c11,  covXZ_h = T_h - m_h (u1'D U'X),  t1 = G~'D u1
            // (zsum then holds u1'D U'X and `colsum` the RAW column sums, d counts U'X only)
};

// one workgroup: column sums, polymorphic flags, T = G'DX, covXZ   (xz: V x d row-major; colsum: V)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(256) void cov_prepare_kernel(const GeneDesc* __restrict__ genes, CovConsts cc,
                                                          double* __restrict__ xz, double* __restrict__ colsum,
                                                          int* __restrict__ poly, double* __restrict__ ustat,
                                                          double* __restrict__ afout) {
  const GeneDesc gd = genes[0];
  const int V = gd.M, d = cc.d;
  for (int h = threadIdx.x; h < V; h += blockDim.x) {
    double s = 0.0, mn = INFINITY, mx = -INFINITY;
    for (int p = 0; p < gd.n_wparts; ++p) {
      const double* c = gd.colstat + (long long)p * 3 * gd.Mp;
      s += c[h];
      mn = fmin(mn, c[gd.Mp + h]);
      mx = fmax(mx, c[2 * gd.Mp + h]);
    }
    if (!cc.fam) {  // family mode: raw column sums and flags were computed before the rotation
      colsum[h] = s;
      poly[h] = (mn == mx) ? 0 : 1;
    } else {
      s = colsum[h];
      if (ustat) {  // family burden tests: score U of the centred column and the GLS allele frequency
        double tu = 0.0, ta = 0.0;
        for (int p = 0; p < gd.n_wparts; ++p) {
          const double* row = gd.parts + (long long)p * gd.Mp * gd.Cp + (long long)h * gd.Cp + V;
          ta += row[d + 1];
          tu += row[d + 2];
        }
        ustat[h] = tu - s * cc.inv_n * cc.k1r;
        afout[h] = (cc.af_denom == 0.0) ? 0.0 : 0.5 * (ta / cc.af_denom);
      }
    }
    for (int k = 0; k < d; ++k) {
      double t = 0.0;
      for (int p = 0; p < gd.n_wparts; ++p)
        t += gd.parts[(long long)p * gd.Mp * gd.Cp + (long long)h * gd.Cp + V + k];
      if (cc.fam)
        xz[(long long)h * d + k] = t - s * cc.inv_n * cc.zsum[k];
      else
        xz[(long long)h * d + k] = cc.binary ? t : (t - s * cc.inv_n * cc.zsum[k]) * cc.inv_sigma2;
    }
  }
}
#endif  // RVT_K_ENGINE

// grid = V workgroups (one per head h): value(h, j) for j >= h into cov[h + j*V]
#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(256) void cov_rows_kernel(const GeneDesc* __restrict__ genes, CovConsts cc,
                                                       const double* __restrict__ xz,
                                                       const double* __restrict__ colsum, double* __restrict__ cov) {
  const GeneDesc gd = genes[0];
  const int V = gd.M, d = cc.d, h = blockIdx.x;
  __shared__ double a[RVT_MAX_COV];  // covXZ_h' covZZInv
  if (threadIdx.x < d) {
    double t = 0.0;
    for (int k = 0; k < d; ++k) t += xz[(long long)h * d + k] * cc.zzinv[k * d + threadIdx.x];
    a[threadIdx.x] = t;
  }
  __syncthreads();
  const double sh = colsum[h];
  __shared__ double t1h;
  if (cc.fam && threadIdx.x == 0) {
    double t = 0.0;
    for (int p = 0; p < gd.n_wparts; ++p) t += gd.parts[(long long)p * gd.Mp * gd.Cp + (long long)h * gd.Cp + V + d];
    t1h = t;
  }
  __syncthreads();
  for (int j = h + threadIdx.x; j < V; j += blockDim.x) {
    double sxx = 0.0;
    for (int p = 0; p < gd.n_wparts; ++p) sxx += gd.parts[(long long)p * gd.Mp * gd.Cp + (long long)h * gd.Cp + j];
    double xx;
    if (cc.fam) {
      double t1j = 0.0;
      for (int p = 0; p < gd.n_wparts; ++p)
        t1j += gd.parts[(long long)p * gd.Mp * gd.Cp + (long long)j * gd.Cp + V + d];
      const double mh = sh * cc.inv_n, mj = colsum[j] * cc.inv_n;
      xx = sxx - mh * t1j - mj * t1h + mh * mj * cc.c11;
    } else {
      xx = cc.binary ? sxx : (sxx - sh * colsum[j] * cc.inv_n) * cc.inv_sigma2;
    }
    double quad = 0.0;
    for (int k = 0; k < d; ++k) quad += a[k] * xz[(long long)j * d + k];
    cov[h + (long long)j * V] = xx - quad;
  }
}
#endif  // RVT_K_ENGINE

// ---- AnalyticVT (src/Model.h:2105-2259, UNRELATED, quantitative trait) from the assembled statistics ---------------------------
// The reference residualises genotypes and phenotype on the covariates and forms u = x'y, v = x'x sigma2
// (:2166-2188): with the intercept in X that is u = G'res and v = (G'G - G'X (X'X)^-1 X'G) sigma2 on the flipped,
// polymorphic columns — exactly the u and Wm that gene_assemble leaves in the gene's scratch.  MultivariateVT::compute
// (regression/MultivariateVT.cpp:22-144): maf = min(af, 1 - af) (af of filtered position a = counter of unfiltered
// column a, as for the SKAT weights); a variant is skipped when maf < 1e-10 or v_aa < 1e-10; the distinct values of
// ceil(maf 1e6) in ascending order are the cutoffs (as doubles: k / 1e6); phi(a, j) = maf_a <= cutoff_j; u_phi = u'phi,
// v_phi = phi' v phi; Stat = max_j |u_phi_j / sqrt(v_phi_jj)| (first maximum); Pvalue = 1 - P(|Z_j| < Stat for all j),
// Z ~ N(0, cor(v_phi)) — rvt_mvn.h.  One 256-thread workgroup per gene.
// vt_mem (doubles): hdr[16 + kMvnShifts] | maf[Mp] | key[Mp] | ord[Mp] | cidx[Mp] | cut[K] | uphi[K] | start[K + 1] | alpha[K] |
//                   A[K x K] | y[kMvnShifts x 256 x K]
// hdr: [0] state (0 nothing to integrate, 1 integrate), [1] K, [2] T, [3] 1 = second integration stage wanted,
//      [16 + j] running sum of the integrand over the points of shift j
constexpr int kVtHdr = 16 + kMvnShifts;
constexpr int kVtStage0 = 4096;  // lattice points per shift of the first integration stage
RVT_HD size_t gene_vt_doubles(int Mp) {
  const size_t K = (size_t)(Mp < kMvnMaxDim ? Mp : kMvnMaxDim);
  return kVtHdr + 4 * (size_t)Mp + 4 * K + 8 + K * K + (size_t)kMvnShifts * 256 * K;
}

// m: columns (flipped, polymorphic); afv[a]: frequency of position a; uvec[a]: score; Wm: m x m (column-major); the
// variance matrix is Wm * sigma2; vt_mem: gene_vt_doubles(Mp) doubles with Mp >= m a multiple of 16.  256 threads.
__device__ void vt_core(int m, int Mp, const double* __restrict__ afv, const double* __restrict__ uvec,
                        const double* __restrict__ Wm, double sigma2, bool binary, double* __restrict__ vt_mem,
                        rvt_gene_result* out) {
  const int tid = threadIdx.x;
  __shared__ int sK, sFail, sMaxIdx;
  __shared__ double sRed[256];
  __shared__ int sRedI[256];
  if (tid == 0) {
    out->vt_ok = 0;
    out->vt_optnum = 0;
    out->vt_ncutoff = 0;
    out->vt_minmaf = out->vt_maxmaf = out->vt_optmaf = out->vt_U = out->vt_V = out->vt_stat = 0.0;
    out->vt_p = out->vt_p_error = 0.0;
    sK = 0;
    sFail = 0;
    if (vt_mem)
      for (int j = 0; j < kVtHdr; ++j) vt_mem[j] = 0.0;
  }
  __syncthreads();
  if (m == 0 || binary || !vt_mem) return;  // "Analytic VT test does not support binary outcomes" (:2143-2149)
  const int Kmax = Mp < kMvnMaxDim ? Mp : kMvnMaxDim;
  double* hdr = vt_mem;
  double* maf = vt_mem + kVtHdr;
  double* keyd = maf + Mp;    // ceil(maf 1e6) as a double, -1: skipped
  double* ordd = keyd + Mp;   // variants ordered by (first cutoff, index)
  double* cidxd = ordd + Mp;  // first cutoff that includes the variant (K: none)
  double* cut = cidxd + Mp;
  double* uphi = cut + Kmax;
  double* startd = uphi + Kmax;  // K + 1 range starts into ord
  double* alpha = startd + Kmax + 8;
  double* A = alpha + Kmax;
  double* ymem = A + (size_t)Kmax * Kmax;
  // ---- 1. maf, skip rule, keys ----------------------------------------------------------------------------------------
  double mn = INFINITY, mx = -INFINITY;
  for (int a = tid; a < m; a += 256) {
    const double f = afv[a];
    const double mf = f < 0.5 ? f : 1.0 - f;
    maf[a] = mf;
    mn = fmin(mn, mf);
    mx = fmax(mx, mf);
    const double vaa = Wm[(size_t)a * m + a] * sigma2;
    keyd[a] = (mf < 1e-10 || vaa < 1e-10) ? -1.0 : (double)(int)ceil(mf * 1000000);
  }
  sRed[tid] = mn;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) sRed[tid] = fmin(sRed[tid], sRed[tid + w]);
    __syncthreads();
  }
  const double minmaf = sRed[0];
  __syncthreads();
  sRed[tid] = mx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) sRed[tid] = fmax(sRed[tid], sRed[tid + w]);
    __syncthreads();
  }
  const double maxmaf = sRed[0];
  __syncthreads();
  // ---- 2. distinct keys in ascending order = cutoffs (a key is a cutoff at its first occurrence) ---------------------------
  for (int a = tid; a < m; a += 256) {
    const double k = keyd[a];
    if (k < 0.0) continue;
    bool first = true;
    int below = 0;  // distinct smaller keys = its position
    for (int b = 0; b < m; ++b) {
      const double kb = keyd[b];
      if (kb < 0.0) continue;
      if (b < a && kb == k) first = false;
      if (kb < k) {
        bool firstb = true;
        for (int c2 = 0; c2 < b; ++c2)
          if (keyd[c2] == kb) {
            firstb = false;
            break;
          }
        if (firstb) ++below;
      }
    }
    if (first) {
      if (below < Kmax) cut[below] = 1.0 * (int)k / 1000000;
      atomicAdd(&sK, 1);
    }
  }
  __syncthreads();
  const int K = sK;
  if (K == 0) return;  // numKeep == 0: compute() returns -1, the row is NA
  if (K > Kmax) {
    if (tid == 0) out->vt_ncutoff = K;
    return;  // (more thresholds than the device integral handles: reported as not fitted)
  }
  // ---- 3. first cutoff of every variant, ordering by it --------------------------------------------------------------------
  for (int a = tid; a < m; a += 256) {
    int ci = K;
    if (keyd[a] >= 0.0) {
      for (int j = 0; j < K; ++j)
        if (maf[a] <= cut[j]) {
          ci = j;
          break;
        }
    }
    cidxd[a] = (double)ci;
  }
  __syncthreads();
  for (int a = tid; a < m; a += 256) {  // rank sort by (cidx, index)
    const double ca = cidxd[a];
    int r = 0;
    for (int b = 0; b < m; ++b) {
      const double cb = cidxd[b];
      if (cb < ca || (cb == ca && b < a)) ++r;
    }
    ordd[r] = (double)a;
  }
  for (int j = tid; j <= K; j += 256) {  // start[j] = number of variants with cidx < j
    int cnt = 0;
    for (int b = 0; b < m; ++b)
      if (cidxd[b] < (double)j) ++cnt;
    startd[j] = (double)cnt;
  }
  __syncthreads();
  // ---- 4. block sums B[p][q] = sum of v over (first cutoff p) x (first cutoff q), then 2-D prefix sums = v_phi ----------------
  for (int e = tid; e < K * K; e += 256) {
    const int p = e / K, q = e % K;
    double sacc = 0.0;
    const int p0 = (int)startd[p], p1 = (int)startd[p + 1], q0 = (int)startd[q], q1 = (int)startd[q + 1];
    for (int x = p0; x < p1; ++x) {
      const int a = (int)ordd[x];
      for (int z = q0; z < q1; ++z) sacc += Wm[(size_t)(int)ordd[z] * m + a];
    }
    A[(size_t)p * K + q] = sacc * sigma2;
  }
  for (int j = tid; j < K; j += 256) {  // u_phi before the prefix: block sums of u
    double sacc = 0.0;
    const int p0 = (int)startd[j], p1 = (int)startd[j + 1];
    for (int x = p0; x < p1; ++x) sacc += uvec[(int)ordd[x]];
    uphi[j] = sacc;
  }
  __syncthreads();
  for (int p = tid; p < K; p += 256)  // prefix along q
    for (int q = 1; q < K; ++q) A[(size_t)p * K + q] += A[(size_t)p * K + q - 1];
  __syncthreads();
  for (int q = tid; q < K; q += 256)  // prefix along p
    for (int p = 1; p < K; ++p) A[(size_t)p * K + q] += A[(size_t)(p - 1) * K + q];
  if (tid == 0)
    for (int j = 1; j < K; ++j) uphi[j] += uphi[j - 1];
  __syncthreads();
  // ---- 5. statistic -----------------------------------------------------------------------------------------------------
  {
    double best = -DBL_MAX;
    int bi = -1;
    for (int j = tid; j < K; j += 256) {
      const double t = fabs(uphi[j] / sqrt(A[(size_t)j * K + j]));
      if (t > best) {
        best = t;
        bi = j;
      }
    }
    sRed[tid] = best;
    sRedI[tid] = bi;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) {  // first maximum: larger value, or equal value with the smaller index
        const double o = sRed[tid + w];
        const int oi = sRedI[tid + w];
        if (oi >= 0 && (sRedI[tid] < 0 || o > sRed[tid] || (o == sRed[tid] && oi < sRedI[tid]))) {
          sRed[tid] = o;
          sRedI[tid] = oi;
        }
      }
      __syncthreads();
    }
    if (tid == 0) sMaxIdx = sRedI[0];
    __syncthreads();
  }
  const int maxIdx = sMaxIdx;
  if (maxIdx < 0) return;  // every statistic is NaN (cannot happen with v_aa >= 1e-10): NA row
  const double T = fabs(uphi[maxIdx] / sqrt(A[(size_t)maxIdx * K + maxIdx]));
  if (tid == 0) {
    out->vt_minmaf = minmaf;
    out->vt_maxmaf = maxmaf;
    out->vt_optmaf = cut[maxIdx];
    out->vt_optnum = (int)startd[maxIdx + 1];
    out->vt_U = uphi[maxIdx];
    out->vt_V = A[(size_t)maxIdx * K + maxIdx];
    out->vt_stat = T;
    out->vt_ncutoff = K;
  }
  if (!(T == T) || !(T < INFINITY)) return;
  if (K == 1) {  // getBandProbFromCor, n == 1 (regression/MultivariateNormalDistribution.cpp:18-21)
    if (tid == 0) {
      out->vt_p = 1.0 - (mvn_phi(T) - mvn_phi(-T));
      out->vt_ok = 1;
    }
    return;
  }
  // ---- 6. correlation, Cholesky factor (columns one after the other, rows in parallel) ------------------------------------------
  for (int j = tid; j < K; j += 256) uphi[j] = sqrt(A[(size_t)j * K + j]);  // uphi is free now: standard deviations
  __syncthreads();
  for (int e = tid; e < K * K; e += 256) {
    const int p = e / K, q = e % K;
    A[e] = (p == q) ? 1.0 : A[e] / (uphi[p] * uphi[q]);
  }
  __syncthreads();
  // Cholesky factor with Genz's variable reordering (mvn_cholesky in rvt_mvn.h, same choices): at step i every thread
  // rates its candidates, the narrowest expected interval wins, rows / columns are swapped, column i is formed
  double* yexp = uphi;  // (free again: conditional expectations of the variables already placed)
  __syncthreads();
  for (int i = 0; i < K; ++i) {
    double bw = 2.0, bv = 0.0, bex = 0.0;
    int bj = -1;
    for (int j = i + tid; j < K; j += 256) {
      double v = A[(size_t)j * K + j], sh = 0.0;
      for (int k = 0; k < i; ++k) {
        v -= A[(size_t)j * K + k] * A[(size_t)j * K + k];
        sh += A[(size_t)j * K + k] * yexp[k];
      }
      double ex;
      const double w = mvn_candidate_width(v, sh, T, 1e-10, &ex);
      if (w < bw) {  // (ascending j inside a thread: the first minimum)
        bw = w;
        bj = j;
        bv = v;
        bex = ex;
      }
    }
    sRed[tid] = bw;
    sRedI[tid] = bj;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) {  // smaller width, or equal width with the smaller index
        const double o = sRed[tid + w];
        const int oi = sRedI[tid + w];
        if (oi >= 0 && (sRedI[tid] < 0 || o < sRed[tid] || (o == sRed[tid] && oi < sRedI[tid]))) {
          sRed[tid] = o;
          sRedI[tid] = oi;
        }
      }
      __syncthreads();
    }
    const int best = sRedI[0];
    __syncthreads();
    if (bj == best) {  // the owner of the winner publishes its variance and expectation
      sRed[0] = bv;
      sRed[1] = bex;
    }
    __syncthreads();
    const double vbest = sRed[0], exbest = sRed[1];
    if (best != i) {
      for (int k = tid; k < K; k += 256) {  // rows
        const double t = A[(size_t)i * K + k];
        A[(size_t)i * K + k] = A[(size_t)best * K + k];
        A[(size_t)best * K + k] = t;
      }
      __syncthreads();
      for (int r = tid; r < K; r += 256) {  // columns
        const double t = A[(size_t)r * K + i];
        A[(size_t)r * K + i] = A[(size_t)r * K + best];
        A[(size_t)r * K + best] = t;
      }
    }
    __syncthreads();
    const double l = vbest > 1e-10 ? sqrt(vbest) : 0.0;
    if (tid == 0) {
      yexp[i] = exbest;
      A[(size_t)i * K + i] = l;
    }
    for (int r = i + 1 + tid; r < K; r += 256) {
      double t = 0.0;
      if (l > 0.0) {
        t = A[(size_t)r * K + i];
        for (int k = 0; k < i; ++k) t -= A[(size_t)r * K + k] * A[(size_t)i * K + k];
        t /= l;
      }
      A[(size_t)r * K + i] = t;
    }
    __syncthreads();
  }
  // ---- 7. lattice generators: frac(sqrt(prime_i)) -----------------------------------------------------------------------------
  if (tid == 0) {
    int found = 0;
    for (int cand = 2; found < K; ++cand) {
      bool prime = true;
      for (int q = 2; q * q <= cand; ++q)
        if (cand % q == 0) {
          prime = false;
          break;
        }
      if (prime) alpha[found++] = mvn_alpha(cand);
    }
  }
  __syncthreads();
  // ---- 8. hand over to the integration kernels (vt_integrate_kernel / vt_finish_kernel) -----------------------------------------------
  if (tid == 0) {
    hdr[1] = (double)K;
    hdr[2] = T;
    __threadfence();
    hdr[0] = 1.0;
  }
}

// vt_mem sections the integration kernels need
__device__ __forceinline__ void vt_sections(double* vt_mem, int Mp, double** alpha, double** A, double** ymem) {
  const int Kmax = Mp < kMvnMaxDim ? Mp : kMvnMaxDim;
  double* cut = vt_mem + kVtHdr + 4 * (size_t)Mp;
  *alpha = cut + 2 * (size_t)Kmax + Kmax + 8;
  *A = *alpha + Kmax;
  *ymem = *A + (size_t)Kmax * Kmax;
}

// The integral.  grid (genes, kMvnShifts), 256 threads: one workgroup sums the integrand over the lattice points of one
// shift — stage 0: points [0, kVtStage0); stage 1 (genes whose first estimate was not accurate enough): the points up to
// kMvnPoints.  One thread = one lattice point at a time; the factor sits in LDS when it fits, the per-thread vectors of
// conditioned values in the workspace ([dimension][thread]: coalesced).
#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(256) void vt_integrate_kernel(const GeneDesc* __restrict__ genes, int stage) {
  const GeneDesc gd = genes[blockIdx.x];
  double* vt_mem = gd.vt_mem;
  if (!vt_mem || vt_mem[0] != 1.0) return;
  if (stage == 1 && vt_mem[3] != 1.0) return;
  const int tid = threadIdx.x, j = blockIdx.y;
  const int K = (int)vt_mem[1];
  const double T = vt_mem[2];
  double *alpha, *A, *ymem;
  vt_sections(vt_mem, gd.Mp, &alpha, &A, &ymem);
  constexpr int kLdsDoubles = 4096;  // factors up to 64 x 64
  __shared__ double sL[kLdsDoubles];
  __shared__ double sRed[256];
  const double* Lp = A;
  if (K * K <= kLdsDoubles) {
    for (int e = tid; e < K * K; e += 256) sL[e] = A[e];
    Lp = sL;
  }
  __syncthreads();
  double* y = ymem + (size_t)j * 256 * K + tid;
  const long long k0 = stage == 0 ? 0 : kVtStage0, k1 = stage == 0 ? kVtStage0 : kMvnPoints;
  double sacc = 0.0;
  for (long long k = k0 + tid; k < k1; k += 256) sacc += mvn_band_point_strided(Lp, K, K, T, alpha, j, k + 1, y, 256);
  sRed[tid] = sacc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) sRed[tid] += sRed[tid + w];
    __syncthreads();
  }
  if (tid == 0) vt_mem[16 + j] += sRed[0];
}
#endif  // RVT_K_ENGINE

// one thread per gene: estimate and error from the shift sums; after stage 0 genes that are not accurate enough ask for
// stage 1
#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ void vt_finish_kernel(const GeneDesc* __restrict__ genes, int n, int stage) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const GeneDesc gd = genes[g];
  double* vt_mem = gd.vt_mem;
  if (!vt_mem || vt_mem[0] != 1.0) return;
  if (stage == 1 && vt_mem[3] != 1.0) return;
  const double P = stage == 0 ? (double)kVtStage0 : (double)kMvnPoints;
  double mean = 0.0, sq = 0.0;
  for (int j = 0; j < kMvnShifts; ++j) {
    const double mj = vt_mem[16 + j] / P;
    mean += mj;
    sq += mj * mj;
  }
  mean /= kMvnShifts;
  const double var = fmax(0.0, sq / kMvnShifts - mean * mean) / (kMvnShifts - 1);
  const double err = 3.5 * sqrt(var);
  rvt_gene_result* out = gd.result;
  out->vt_p = 1.0 - mean;
  out->vt_p_error = err;
  out->vt_ok = 1;
  if (stage == 0) vt_mem[3] = err < 2.5e-4 ? 0.0 : 1.0;  // (the reference asks its rule for 1e-3)
}
#endif  // RVT_K_ENGINE

// the unrelated-sample test: u and Wm are where gene_assemble left them (one workgroup per gene)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(256) void gene_vt_kernel(const GeneDesc* __restrict__ genes,
                                                      const NullConsts* __restrict__ ncp) {
  const GeneDesc gd = genes[blockIdx.x];
  const GeneScratch ws = gene_scratch_carve(gd.scratch, gd.Mp, gd.Cp);
  const int* kidx = ws.ivec;
  const int m = kidx[gd.Mp], M = gd.M, d = ncp->d, ldr = gd.Cp;
  double* uvec = gd.vt_mem ? gd.vt_mem + gene_vt_doubles(gd.Mp) : nullptr;  // Mp doubles behind the core's workspace
  if (uvec)
    for (int a = threadIdx.x; a < m; a += 256) uvec[a] = ws.R[(size_t)kidx[a] * ldr + M + d];
  __syncthreads();
  vt_core(m, gd.Mp, gd.af, uvec, ws.Wm, ncp->sigma2, ncp->binary != 0, gd.vt_mem, gd.result);
}
#endif  // RVT_K_ENGINE

// the related-sample test (FamAnalyticVT): frequency, score and variance matrix prepared by the host from the family
// covariance machinery; buf = af[m] | u[m] | V[m x m] | workspace
#if !defined(RVT_K_SPLIT) || defined(RVT_K_FAM)
static __global__ __launch_bounds__(256) void vt_direct_kernel(int m, int Mp, double* __restrict__ buf, rvt_gene_result* out) {
  vt_core(m, Mp, buf, buf + m, buf + 2 * (size_t)m, 1.0, false, buf + 2 * (size_t)m + (size_t)m * m, out);
}
#endif  // RVT_K_FAM

// ---- MetaScoreTest, unrelated samples: single-variant score statistics of one block from the same partials ----
//   quantitative (MetaUnrelatedQtl, src/Model.h:3516-3549 over LinearRegressionScoreTest.cpp:173-263):
//     U = g'res, SS = g'g - g'X (X'X)^-1 X'g;  U_STAT = U / sigma2, V_STAT = SS / sigma2, effect = U / SS,
//     SE = sigma2 / sqrt(SS sigma2), stat = U^2 / (SS sigma2)
//   binary (MetaUnrelatedBinary, src/Model.h:3706-3769 over LogisticRegressionScoreTest.cpp:220-302):
//     U = g'(y - p), V = g'Wg - g'WX (X'WX)^-1 X'Wg, effect = U / V (0 when U == 0), SE = 1 / sqrt(V), stat = U^2 / V
//     (for d > 1 the reference's llt().solve(Identity(d, d)) on the 1 x 1 SS is a dimension mismatch, SURVEY quirk
//     #15; the intended 1-df statistic is returned, as for CMC / Zeggini)
// The block is submitted as one "gene" per slice of <= 16 columns (only the diagonal tile and the [X | rr] tile are
// needed, so each slice streams once at the narrow-class rate); gene_id carries the slice's first column.
// out: ustat | vstat | effect | se | pval (vt entries each); ok[h] = 1 when the site is polymorphic and SS > 0.
#if !defined(RVT_K_SPLIT) || defined(RVT_K_ENGINE)
static __global__ __launch_bounds__(64) void score_finish_kernel(const GeneDesc* __restrict__ genes,
                                                          const NullConsts* __restrict__ ncp, int vt,
                                                          double* __restrict__ out, int* __restrict__ ok) {
  const GeneDesc gd = genes[blockIdx.x];
  const int V = gd.M, h = threadIdx.x;
  if (h >= V) return;
  const long long col = gd.gene_id + h;
  double mn = INFINITY, mx = -INFINITY, cm = 0.0;
  unsigned long long orb = 0ull;
  // a hard-call slice that was handed back (gene_flags_hc_kernel) holds the general kernel's statistics
  const bool hcs = gd.hc && gd.flags[2 * gd.MT + 1] == 0;
  const int cs_rows = hcs ? kHcColstatRows : 3;
  for (int p = 0; p < gd.n_wparts; ++p) {
    const double* c = gd.colstat + (long long)p * cs_rows * gd.Mp;
    mn = fmin(mn, c[gd.Mp + h]);
    mx = fmax(mx, c[2 * gd.Mp + h]);
    if (hcs) {
      cm += c[3 * gd.Mp + h];
      orb |= reinterpret_cast<const unsigned long long*>(c)[4 * gd.Mp + h];
    }
  }
  const double mu = rvt_bits_to_double(orb);
  if (cm > 0.0) {  // mean-imputed column (suffstat_hc.hip.h): one value for every masked entry
    mn = fmin(mn, mu);
    mx = fmax(mx, mu);
  }
  const int polymorphic = (mn == mx) ? 0 : 1;
  const int d = ncp->d, binary = ncp->binary;
  const double sigma2 = ncp->sigma2;
  double shh = 0.0, u = 0.0, t[RVT_MAX_COV];
  for (int k = 0; k < d; ++k) t[k] = 0.0;
  for (int p = 0; p < gd.n_wparts; ++p) {
    const double* row = gd.parts + (long long)p * gd.Mp * gd.Cp + (long long)h * gd.Cp;
    shh += row[h];
    for (int k = 0; k < d; ++k) t[k] += row[V + k];
    u += row[V + d];
  }
  // the diagonal of the hard-call Gram tile is sum (H + 4m)^2 = sum H^2 + 16 cm, and H = 0 where masked
  if (cm > 0.0) shh = (shh - 16.0 * cm) + (mu * mu) * cm;
  double q = 0.0;
  for (int k = 0; k < d; ++k) {
    double s = 0.0;
    for (int l = 0; l < d; ++l) s += ncp->Cinv[k * d + l] * t[l];
    q += t[k] * s;
  }
  const double SS = shh - q;
  const int fit = polymorphic && SS > 0.0;
  double us = 0.0, vs = 0.0, eff = 0.0, se = 0.0, pv = 1.0;
  if (fit) {
    if (!binary) {
      us = u / sigma2;
      vs = SS / sigma2;
      eff = u / SS;
      se = sigma2 / sqrt(SS * sigma2);
      double SSi = 1.0 / SS;
      SSi /= sigma2;
      pv = chisq_Q(u * SSi * u, 1.0);
    } else {
      us = u;
      vs = SS;
      eff = (u != 0.0) ? u / SS : 0.0;
      se = 1.0 / sqrt(SS);
      pv = chisq_Q(u * (1.0 / SS) * u, 1.0);
    }
  }
  out[col] = us;
  out[(long long)vt + col] = vs;
  out[2LL * vt + col] = eff;
  out[3LL * vt + col] = se;
  out[4LL * vt + col] = pv;
  ok[col] = fit;
}
#endif  // RVT_K_ENGINE

// ---- MetaCov for windows wider than one block: heads x window rectangle from two plain GEMMs ------------------
// S = G_H' D G_W (H x W, column-major) and T = G_W' D X (W x d, column-major) come from the integer-plane products of
// rot_gemm.hip.h (gemm_tn_planes; exact for hard calls and an unweighted model); cs = raw column sums
// of the W window columns (the H heads are its first H columns).  Unrelated samples only.
#if !defined(RVT_K_SPLIT) || defined(RVT_K_META)
static __global__ void cov_rect_xz_kernel(CovConsts cc, const double* __restrict__ T, const double* __restrict__ cs, int W,
                                   double* __restrict__ xz) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= W) return;
  for (int k = 0; k < cc.d; ++k) {
    const double t = T[j + (long long)k * W];
    xz[(long long)j * cc.d + k] = cc.binary ? t : (t - cs[j] * cc.inv_n * cc.zsum[k]) * cc.inv_sigma2;
  }
}
#endif  // RVT_K_META

// grid = H workgroups: cov[h + j*H] for j >= h
#if !defined(RVT_K_SPLIT) || defined(RVT_K_META)
static __global__ __launch_bounds__(256) void cov_rect_rows_kernel(CovConsts cc, const double* __restrict__ S,
                                                            const double* __restrict__ cs,
                                                            const double* __restrict__ xz, int H, int W,
                                                            double* __restrict__ cov) {
  const int h = blockIdx.x, d = cc.d;
  __shared__ double a[RVT_MAX_COV];
  if (threadIdx.x < d) {
    double t = 0.0;
    for (int k = 0; k < d; ++k) t += xz[(long long)h * d + k] * cc.zzinv[k * d + threadIdx.x];
    a[threadIdx.x] = t;
  }
  __syncthreads();
  const double sh = cs[h];
  for (int j = h + threadIdx.x; j < W; j += blockDim.x) {
    const double sxx = S[h + (long long)j * H];
    const double xx = cc.binary ? sxx : (sxx - sh * cs[j] * cc.inv_n) * cc.inv_sigma2;
    double quad = 0.0;
    for (int k = 0; k < d; ++k) quad += a[k] * xz[(long long)j * d + k];
    cov[h + (long long)j * H] = xx - quad;
  }
}
#endif  // RVT_K_META

// family mode of the two rectangle kernels (MetaCovFamQtl on rotated data, regression/FastLMM.cpp:510-595): T holds
// G~_W' D [U'X | u1] (W x (d + 1), column-major), cs the RAW column sums; see CovConsts for the centring algebra
#if !defined(RVT_K_SPLIT) || defined(RVT_K_META)
static __global__ void cov_rect_fam_xz_kernel(CovConsts cc, const double* __restrict__ T, const double* __restrict__ cs, int W,
                                       double* __restrict__ xz, double* __restrict__ t1) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= W) return;
  const double m = cs[j] * cc.inv_n;
  for (int k = 0; k < cc.d; ++k) xz[(long long)j * cc.d + k] = T[j + (long long)k * W] - m * cc.zsum[k];
  t1[j] = T[j + (long long)cc.d * W];
}
#endif  // RVT_K_META
#if !defined(RVT_K_SPLIT) || defined(RVT_K_META)
static __global__ __launch_bounds__(256) void cov_rect_fam_rows_kernel(CovConsts cc, const double* __restrict__ S,
                                                                const double* __restrict__ cs,
                                                                const double* __restrict__ xz,
                                                                const double* __restrict__ t1, int H, int W,
                                                                double* __restrict__ cov) {
  const int h = blockIdx.x, d = cc.d;
  __shared__ double a[RVT_MAX_COV];
  if (threadIdx.x < d) {
    double t = 0.0;
    for (int k = 0; k < d; ++k) t += xz[(long long)h * d + k] * cc.zzinv[k * d + threadIdx.x];
    a[threadIdx.x] = t;
  }
  __syncthreads();
  const double mh = cs[h] * cc.inv_n, t1h = t1[h];
  for (int j = h + threadIdx.x; j < W; j += blockDim.x) {
    const double mj = cs[j] * cc.inv_n;
    const double xx = S[h + (long long)j * H] - mh * t1[j] - mj * t1h + mh * mj * cc.c11;
    double quad = 0.0;
    for (int k = 0; k < d; ++k) quad += a[k] * xz[(long long)j * d + k];
    cov[h + (long long)j * H] = xx - quad;
  }
}
#endif  // RVT_K_META

// MetaCov fast path for hard-call blocks: ONE pass over the window's columns produces everything the band needs
// besides G'G — raw column sums and polymorphic flags (as raw_colstat_kernel), T = G'X (W x d, column-major) and the int8
// copy of the columns that the exact integer product G'G reads (rot_gemm.hip.h).  Four columns per workgroup share the
// loads of X; the rows are cut into gridDim.y slices whose partial results (part[slice][column][3 + DMAX]: sum, min,
// max, T) are added in a fixed order by cov_hc_finish_kernel.  grid = (ceil(W / 4), slices), 256 threads; d <= DMAX.
// PACK = false (round 5): the same pass for blocks that are NOT hard calls (the fp64 band, gemm_f64.hip.h) — no int8 copy,
// no content test, and optional weights wts (a binary trait's p(1 - p)): T = G' diag(wts) X.
constexpr int kCovHcCols = 4;
template <int DMAX, bool PACK = true>
__global__ __launch_bounds__(256) void cov_hc_prep_kernel(const double* __restrict__ G, long long N, long long ld, int W,
                                                          const double* __restrict__ X, long long ldx, int d,
                                                          signed char* __restrict__ out8, long long ldk,
                                                          double* __restrict__ part, int* __restrict__ bad,
                                                          const double* __restrict__ wts = nullptr,
                                                          int* __restrict__ hard_flag = nullptr, int ring = 0,
                                                          int ring_col0 = 0, unsigned char* __restrict__ out4 = nullptr,
                                                          long long ldk4 = 0, const double* __restrict__ mu_known = nullptr,
                                                          unsigned char* __restrict__ out4m = nullptr) {
  // out4 (PACK only, optional): the same hard calls as 4-bit E2M1 codes (0 -> 0x0, 1 -> 0x2, 2 -> 0x4), sample 2 i in the low
  // nibble of byte i — what the MXFP4 band product reads (band_gemm.hip.h); out8 may then be null
  // mu_known / out4m (PACK only, optional; round 6): column j is hard calls plus ONE other value mu_known[j] (the mean that
  // consolidate() imputed; NaN = none) — known because the column crossed PCIe as 2-bit codes.  Then out4 holds the hard-call
  // part h (0 where the entry is mu) and out4m the mask m (code of 1 where the entry is mu): g = h + mu m, and the band of such
  // columns is four exact integer products (h'h, h'm, m'h, m'm) combined with the mu's in fp64 (band_finish_i32_kernel)
  // ring > 0: G is the base of a block used as a ring of `ring` columns, column j of the call is the physical column
  // (ring_col0 + j) mod ring (MetaCov's circular window); the outputs (out8, part) are indexed by j
  const int c0 = blockIdx.x * kCovHcCols;
  bool not_hard = false;  // a value other than 0.0 / 1.0 / 2.0: the int8 copy is not the block (the host falls back)
  const int nc = min(kCovHcCols, W - c0);
  // Round 5: a lane takes FOUR consecutive samples per column and step — two 16-byte loads, one 4-byte store of the packed
  // int8 copy (the first version loaded 8 bytes and stored single bytes per lane: 2.6 TB/s on a pass that only streams).
  // The block's pad rows (N .. ld, ld a multiple of 16) are zero and readable: only min / max have to skip them.
  const long long per = ((N + gridDim.y - 1) / gridDim.y + 1023) / 1024 * 1024;
  const long long i0 = (long long)blockIdx.y * per, i1 = (i0 + per < N) ? i0 + per : N;
  double s[kCovHcCols], mn[kCovHcCols], mx[kCovHcCols], t[kCovHcCols][DMAX];
#pragma unroll
  for (int c = 0; c < kCovHcCols; ++c) {
    s[c] = 0.0;
    mn[c] = INFINITY;
    mx[c] = -INFINITY;
#pragma unroll
    for (int k = 0; k < DMAX; ++k) t[c][k] = 0.0;
  }
  for (long long i = i0 + 4 * (long long)threadIdx.x; i < i1; i += 1024) {
    double x[DMAX][4];
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
      if (k < d) {
        const double2 a = *reinterpret_cast<const double2*>(X + (long long)k * ldx + i);
        const double2 b = *reinterpret_cast<const double2*>(X + (long long)k * ldx + i + 2);
        x[k][0] = a.x;
        x[k][1] = a.y;
        x[k][2] = b.x;
        x[k][3] = b.y;
      } else {
        x[k][0] = x[k][1] = x[k][2] = x[k][3] = 0.0;
      }
    }
    if (!PACK && wts) {
      const double2 a = *reinterpret_cast<const double2*>(wts + i);
      const double2 b = *reinterpret_cast<const double2*>(wts + i + 2);
#pragma unroll
      for (int k = 0; k < DMAX; ++k) {
        x[k][0] *= a.x;
        x[k][1] *= a.y;
        x[k][2] *= b.x;
        x[k][3] *= b.y;
      }
    }
    const int live = (int)((i1 - i < 4) ? i1 - i : 4);  // samples of this group inside the slice (the rest: pad rows, zero)
#pragma unroll
    for (int c = 0; c < kCovHcCols; ++c) {
      if (c < nc) {
        long long pc = c0 + c;
        if (ring > 0) {
          pc += ring_col0;
          if (pc >= ring) pc -= ring;
        }
        const double* gp = G + pc * ld + i;
        const double2 a = *reinterpret_cast<const double2*>(gp);
        const double2 b = *reinterpret_cast<const double2*>(gp + 2);
        const double g[4] = {a.x, a.y, b.x, b.y};
        unsigned packed = 0, hpacked = 0, mpacked = 0;
        const double muc = (PACK && mu_known) ? mu_known[c0 + c] : NAN;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (PACK) {
            const bool hard = g[e] == 0.0 || g[e] == 1.0 || g[e] == 2.0;
            const bool is_mu = !hard && g[e] == muc;  // (NaN compares false: a column without an other value)
            packed |= ((unsigned)(int)g[e] & 0xffu) << (8 * e);
            hpacked |= (hard ? ((unsigned)(int)g[e] & 0xffu) : 0u) << (8 * e);
            mpacked |= (is_mu ? 1u : 0u) << (8 * e);
            not_hard |= !(hard || is_mu);
          }
          s[c] += g[e];
          if (e < live) {
            mn[c] = fmin(mn[c], g[e]);
            mx[c] = fmax(mx[c], g[e]);
          }
#pragma unroll
          for (int k = 0; k < DMAX; ++k) t[c][k] = fma(g[e], x[k][e], t[c][k]);
        }
        if (PACK && out8) *reinterpret_cast<unsigned*>(out8 + (long long)(c0 + c) * ldk + i) = packed;
        if (PACK && out4) {
          // bytes 0..3 of `hpacked` hold the hard-call part in {0, 1, 2}: code = value << 1, two codes per byte
          const unsigned q = (hpacked << 1) & 0x0e0e0e0eu;
          *reinterpret_cast<unsigned short*>(out4 + (long long)(c0 + c) * ldk4 + (i >> 1)) =
              (unsigned short)((q & 0xfu) | ((q >> 4) & 0xf0u) | ((q >> 8) & 0xf00u) | ((q >> 12) & 0xf000u));
        }
        if (PACK && out4m) {
          const unsigned q = (mpacked << 1) & 0x0e0e0e0eu;
          *reinterpret_cast<unsigned short*>(out4m + (long long)(c0 + c) * ldk4 + (i >> 1)) =
              (unsigned short)((q & 0xfu) | ((q >> 4) & 0xf0u) | ((q >> 8) & 0xf00u) | ((q >> 12) & 0xf000u));
        }
      }
    }
  }
  __shared__ double red[4][kCovHcCols][DMAX + 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (PACK && __any(not_hard) && lane == 0) {
    if (bad) atomicOr(bad, 1);
    if (hard_flag) atomicAnd(hard_flag, 0);  // (the per-column flag of rvt_block_upload_columns: nonzero = hard calls only)
  }
#pragma unroll
  for (int c = 0; c < kCovHcCols; ++c) {
    double v = s[c], a = mn[c], b = mx[c];
    for (int o = 32; o > 0; o >>= 1) {
      v += __shfl_down(v, o);
      a = fmin(a, __shfl_down(a, o));
      b = fmax(b, __shfl_down(b, o));
    }
    if (lane == 0) {
      red[wave][c][0] = v;
      red[wave][c][1] = a;
      red[wave][c][2] = b;
    }
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
      double u = t[c][k];
      for (int o = 32; o > 0; o >>= 1) u += __shfl_down(u, o);
      if (lane == 0) red[wave][c][3 + k] = u;
    }
  }
  __syncthreads();
  if (threadIdx.x < kCovHcCols * (DMAX + 3)) {
    const int c = threadIdx.x / (DMAX + 3), f = threadIdx.x % (DMAX + 3);
    if (c < nc) {
      double r;
      if (f == 1)
        r = fmin(fmin(red[0][c][1], red[1][c][1]), fmin(red[2][c][1], red[3][c][1]));
      else if (f == 2)
        r = fmax(fmax(red[0][c][2], red[1][c][2]), fmax(red[2][c][2], red[3][c][2]));
      else
        r = red[0][c][f] + red[1][c][f] + red[2][c][f] + red[3][c][f];
      part[((long long)blockIdx.y * W + c0 + c) * (DMAX + 3) + f] = r;
    }
  }
}

// one thread per (column, field): colsum, poly, T (W x d column-major)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_META)
// (t_row_stride > 0: T row-major, T[j * t_row_stride + k] — the layout of a block's column cache — instead of T[j + k W])
static __global__ void cov_hc_finish_kernel(const double* __restrict__ part, int slices, int W, int d, int dmax,
                                     double* __restrict__ colsum, int* __restrict__ poly, double* __restrict__ T,
                                     int t_row_stride = 0) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int F = dmax + 3;
  if (idx >= W * F) return;
  const int j = idx / F, f = idx % F;
  if (f == 2 || f - 3 >= d) return;  // (max is folded into f == 1)
  if (f == 1) {
    double a = INFINITY, b = -INFINITY;
    for (int s = 0; s < slices; ++s) {
      a = fmin(a, part[((long long)s * W + j) * F + 1]);
      b = fmax(b, part[((long long)s * W + j) * F + 2]);
    }
    poly[j] = (a != b) ? 1 : 0;
    return;
  }
  double r = 0.0;
  for (int s = 0; s < slices; ++s) r += part[((long long)s * W + j) * F + f];
  if (f == 0)
    colsum[j] = r;
  else if (t_row_stride > 0)
    T[(long long)j * t_row_stride + (f - 3)] = r;
  else
    T[j + (long long)(f - 3) * W] = r;
}
#endif  // RVT_K_META

// the column statistics a block keeps per column (rvt_ctx::ColKind) -> the work arrays of a covariance call: W columns from col0
#if !defined(RVT_K_SPLIT) || defined(RVT_K_META)
static __global__ void cov_cache_gather_kernel(const double* __restrict__ cs_c, const int* __restrict__ poly_c,
                                        const double* __restrict__ T_c, int W, int d, int t_stride, double* __restrict__ colsum,
                                        int* __restrict__ poly, double* __restrict__ T) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= W) return;
  colsum[j] = cs_c[j];
  poly[j] = poly_c[j];
  for (int k = 0; k < d; ++k) T[j + (long long)k * W] = T_c[(long long)j * t_stride + k];
}
#endif  // RVT_K_META

// dst[i + k*ld] = src[i + k*ld] * v[i]  (binary trait: one GEMM operand carries the weights)
#if !defined(RVT_K_SPLIT) || defined(RVT_K_META)
static __global__ void scale_rows_kernel(const double* __restrict__ src, const double* __restrict__ v, long long N,
                                  long long ld, double* __restrict__ dst) {
  const double* s = src + (long long)blockIdx.y * ld;
  double* d = dst + (long long)blockIdx.y * ld;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x)
    d[i] = s[i] * v[i];
}
#endif  // RVT_K_META

// =====================================================================================================
// K4: p-values, one wave per gene.
//
// Lane t owns quadrature abscissa t (21 for the first panel, 42 for the two halves of a bisected interval);
// lanes 0..10 also do the per-rho tails and quantiles; lane 0 runs the QAGS bookkeeping.  The expensive
// part of a Davies evaluation is its main integration — (nt+1) terms, each a loop over all coefficients —
// and in one QAGS step only the abscissae with Q >= 0 need it, so the lanes would mostly idle if each
// integrated its own point.  Instead every lane runs qf() up to the integration (davies_qf_front), the
// (point, term) pairs of ALL lanes are laid out in one flat list, the 64 lanes evaluate 64 terms at a time
// (davies_term), and each point's lane then adds its terms in the reference's order (k = nt .. 0), so the
// sums are bit-identical to the sequential loop of qfc.c:241-270.
// =====================================================================================================
constexpr int kTermCap = 256;  // (point, term) pairs staged in LDS per pass

struct WaveDaviesLds {  // (LDS pointers by type: see RVT_LDSQ in rvt_davies.h)
  RVT_LDSQ double* c;      // [64]
  RVT_LDSQ double* sig;    // [64]
  RVT_LDSQ double* intv;   // [64]
  RVT_LDSQ int* nt1;       // [64]  nt+1, 0 = no main integration
  RVT_LDSQ int* which;     // [64]  coefficient set of the lane's task (bit 0), bit 1: none of its coefficients is negative
  RVT_LDSQ int* off;       // [65]  exclusive prefix of nt1
  RVT_LDSQ double* v1;     // [kTermCap]
  RVT_LDSQ double* v2;     // [kTermCap]
};
__device__ __forceinline__ size_t wave_davies_lds_bytes() {
  return sizeof(double) * (3 * 64 + 2 * kTermCap) + sizeof(int) * (64 + 64 + 66);
}
__device__ __forceinline__ WaveDaviesLds wave_davies_lds_carve(char* mem) {
  WaveDaviesLds w;
  RVT_LDSQ double* d = (RVT_LDSQ double*)reinterpret_cast<double*>(mem);
  w.c = d;
  w.sig = d + 64;
  w.intv = d + 128;
  w.v1 = d + 192;
  w.v2 = w.v1 + kTermCap;
  RVT_LDSQ int* ip = (RVT_LDSQ int*)(w.v2 + kTermCap);
  w.nt1 = ip;
  w.which = ip + 64;
  w.off = ip + 128;
  return w;
}

// MixtureChiSquare::getPvalue for one point per lane (regression/MixtureChiSquare.cpp:7-29), cooperative
// main integration.  Every lane of the wave must call it; `active` says whether the lane has a point.
// lbs/ths/rs: the (at most two) coefficient sets in LDS; `which` selects the lane's set.
#ifdef RVT_PROF_K4
#define RVT_K4_TICK(slot, t0) do { if (prof4) prof4[slot] += (double)(clock64() - (t0)); } while (0)
#else
#define RVT_K4_TICK(slot, t0) do { } while (0)
#endif
template <bool FAST>
__device__ __forceinline__ double wave_davies_pvalue(bool active, int which, const dv_coefs* lbs,
                                                     const dv_coefs* lss, const int* rs, double c,
                                                     dv_pre_cp pre, dv_memo_p memo, int lane,
                                                     const WaveDaviesLds& L, double* nterms, double* prof4 = nullptr) {
#ifdef RVT_PROF_K4
  const long long tq0 = clock64();
#endif
  DaviesTask task;
  task.need_main = false;
  task.fault = 0;
  task.qfval = 0.0;
  task.nterms = 0.0;
  task.intl = task.ersm = 0.0;
  task.acc = 0.000001;
  task.c = task.sigsq = task.intv = 0.0;
  task.nt = 0;
  task.allpos = false;
  bool direct = true;   // result already known (Liu for a single coefficient, 1.0 for c < 0, or inactive)
  double pdirect = 0.0;
  if (active) {
    const dv_coefs lb = which ? lbs[1] : lbs[0];
    const int r = rs[which];
    if (r == 1) {
      const double l1 = lb[0];
      pdirect = liu_pvalue(&l1, r, c);
    } else if (c < 0.0) {
      pdirect = 1.0;  // see davies_pvalue(): qf() = 0 or a fault, both replaced by Liu at every call site
    } else {
      direct = false;
      davies_qf_front_t<FAST>(lb, nullptr, r, c, 10000, 0.000001, pre, &task, memo, which ? lss[1] : lss[0]);
    }
  }

  RVT_K4_TICK(0, tq0);  // front (per-lane searches)
#ifdef RVT_PROF_K4
  const long long tq1 = clock64();
#endif
  const bool need = active && !direct && task.need_main;
  L.nt1[lane] = need ? task.nt + 1 : 0;
  L.which[lane] = which | (task.allpos ? 2 : 0);
  L.c[lane] = task.c;
  L.sig[lane] = task.sigsq;
  L.intv[lane] = task.intv;
  __syncthreads();
  if (lane == 0) {
    int acc = 0;
    for (int q = 0; q < 64; ++q) {
      L.off[q] = acc;
      acc += L.nt1[q];
    }
    L.off[64] = acc;
  }
  __syncthreads();
  const int total = L.off[64];
  const int my_begin = L.off[lane], my_end = L.off[lane + 1];
  double intl = task.intl, ersm = task.ersm;
  for (int cs = 0; cs < total; cs += kTermCap) {
    const int ce = (cs + kTermCap < total) ? cs + kTermCap : total;
    for (int idx = cs + lane; idx < ce; idx += 64) {
      int p = 0;  // owner of this flat index: the LAST q with off[q] <= idx (owners with nt1 == 0 share their offset
                  // with the next one; the last of them is the one that really owns idx) — off is non-decreasing
#pragma unroll
      for (int step = 32; step > 0; step >>= 1) p += (L.off[p + step] <= idx) ? step : 0;
      const int k = (L.nt1[p] - 1) - (idx - L.off[p]);
      const int wf = L.which[p], w = wf & 1;
      double t1, t2;
      davies_term_t<FAST>(w ? lbs[1] : lbs[0], w ? rs[1] : rs[0], L.c[p], L.sig[p], L.intv[p], k, &t1, &t2, (wf & 2) != 0);
      L.v1[idx - cs] = t1;
      L.v2[idx - cs] = t2;
    }
    __syncthreads();
    if (need) {
      const int b = (my_begin > cs) ? my_begin : cs, e = (my_end < ce) ? my_end : ce;
      for (int idx = b; idx < e; ++idx) {  // k descending = flat index ascending: the reference's order
        intl = intl + L.v1[idx - cs];
        ersm = ersm + L.v2[idx - cs];
      }
    }
    __syncthreads();
  }
  RVT_K4_TICK(1, tq1);  // flattened main integration
  if (!active) return 0.0;
  if (direct) return pdirect;
  if (need) task.nterms += task.nt + 1;
  *nterms += task.nterms;
  int fault;
  double p = 1.0 - davies_qf_back(task, intl, ersm, &fault);
  if (p > 1.0) p = 1.0;
  if (fault) p = -1.0;
  return p;
}

// Per-gene state of the p-value kernel that every lane reads (or lane 0 updates): in LDS, so that nothing of it is
// replicated per lane in registers / scratch memory (the kernel used to carry 2.5 KB of scratch per lane, i.e. global
// memory round trips inside the sequential QAGS bookkeeping and the per-abscissa integrand set-up).
constexpr int kQagsLds = 64;  // QAGS intervals kept in LDS; a store that outgrows it moves to the global workspace
struct PvShared {
  GeneStats gs;
  SkatoIntegrand si;
  DaviesPrelude pre;       // SKAT-O's coefficients (eigenvalues of Z(I-M)Z')
  DaviesPrelude pre_skat;  // SKAT's own
  DaviesMemo memo, memo_skat;
  LiuPre liu;
  QagsMachine qm;
  double pvals[kNRho];
  double ctl[4];  // a1, b1, b2, running
  double alist[kQagsLds], blist[kQagsLds], rlist[kQagsLds], elist[kQagsLds];
  int order[kQagsLds], level[kQagsLds];
};

// the coefficients by decreasing |lb| (stable): ls[rank(j)] = lb[j] — davies_order() as a rank computation, one element per
// lane (dv_cfe, the only user of the order, reads the coefficients themselves in that order)
__device__ __forceinline__ void wave_davies_sorted(const double* lb, int r, double* ls, int lane) {
  for (int j = lane; j < r; j += 64) {
    const double lj = fabs(lb[j]);
    int rank = 0;
    for (int k = 0; k < r; ++k) {
      const double lk = fabs(lb[k]);
      rank += (lk > lj || (lk == lj && k < j)) ? 1 : 0;
    }
    ls[rank] = lb[j];
  }
}

// FAST = true: product form of Davies' coefficient sums (rvt_davies.h), the default; FAST = false: the term-by-term
// form (RVT_TEST_EXACT_DAVIES).
#ifndef RVT_PV_WAVES
#define RVT_PV_WAVES 2  // waves per SIMD the product-form kernel is compiled for
#endif
template <bool FAST>
__global__ __launch_bounds__(64, FAST ? RVT_PV_WAVES : 2) void gene_pvalue_kernel(const GeneDesc* __restrict__ genes,
                                                                       unsigned tests) {
#ifdef RVT_PROF_K4
  const long long tk_entry = clock64();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ PvShared sh;
  const GeneDesc& gd = genes[genes[blockIdx.x].pv_gene];
  const int lane = threadIdx.x;
  {  // GeneStats -> LDS
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(gd.stats);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&sh.gs);
    for (int i = lane; i < (int)(sizeof(GeneStats) / 8); i += 64) dst[i] = src[i];
  }
  __syncthreads();
  const GeneStats& gs = sh.gs;
  const int M = gd.M;
  const int Me = (M + 1) & ~1;  // (every array starts on a 16-byte boundary)
  double* lam_skat = reinterpret_cast<double*>(smem);
  double* lam_zimz = lam_skat + Me;
  double* ls_skat = lam_zimz + Me;   // the same coefficients by decreasing magnitude (dv_cfe)
  double* ls_zimz = ls_skat + Me;
  double* fv = ls_zimz + Me;         // 42 integrand values
  const WaveDaviesLds L = wave_davies_lds_carve(reinterpret_cast<char*>(fv + 42));
  rvt_gene_result* const out = gd.result;
  const int n_skat = gs.skat_nlambda, n_zimz = gs.zimz_nlambda;
  for (int i = lane; i < n_skat; i += 64) lam_skat[i] = gd.lambda[gs.skat_lambda_off + i];
  for (int i = lane; i < n_zimz; i += 64) lam_zimz[i] = gd.lambda[gs.zimz_lambda_off + i];
  __syncthreads();
  if (lane == 0) pvalue_init_result(gs, gd.gene_id, out);
  if (gs.n_poly == 0) return;
  wave_davies_sorted(lam_skat, n_skat, ls_skat, lane);
  wave_davies_sorted(lam_zimz, n_zimz, ls_zimz, lane);
  dv_memo_clear((dv_memo_p)&sh.memo, lane, 64);
  dv_memo_clear((dv_memo_p)&sh.memo_skat, lane, 64);
  __syncthreads();
  // (address-space-3 pointers: every read of the coefficients in the loops of rvt_davies.h is a ds_read)
  const dv_coefs lbs[2] = {(dv_coefs)lam_zimz, (dv_coefs)lam_skat};
  const dv_coefs lss[2] = {(dv_coefs)ls_zimz, (dv_coefs)ls_skat};
  const int rs[2] = {n_zimz, n_skat};
#ifdef RVT_PROF_K4
  const long long tk_a = clock64();  // loads, sorted copies, memo clear
#endif
  double terms = 0.0;
  const bool fam = (tests & RVT_TEST_FAMSKAT) != 0;  // FamSkat.cpp:118: Davies only, result in the famskat fields
  const bool do_skat = (tests & (RVT_TEST_SKAT | RVT_TEST_FAMSKAT)) != 0;
  const bool do_skato = (tests & RVT_TEST_SKATO) && skato_fit_ok(gs);
  const bool skato_quad = do_skato && !gs.skato_single;
  // ---- per-rho tails (lanes 0..10), burden tails (lanes 60, 61) ------------------------------------------
  double pv_rho = 1.0;
  SkatoMoment mo;
  mo.muQ = mo.varQ = mo.df = 1.0;
  if (lane < kNRho && skato_quad) {
    mo.muQ = gs.mom_mu[lane];
    mo.varQ = gs.mom_var[lane];
    mo.df = gs.mom_df[lane];
    pv_rho = skato_p_by_moment(gs.Qs[lane], mo);
    sh.pvals[lane] = pv_rho;
  } else if (lane == 61 && (tests & RVT_TEST_CMC) && gs.cmc_ok) {
    pv_rho = chisq_Q(gs.cmc_stat, 1.0);
  } else if (lane == 60 && (tests & RVT_TEST_ZEGGINI) && gs.zeg_ok) {
    pv_rho = chisq_Q(gs.zeg_stat, 1.0);
  }
  const double cmc_p = __shfl(pv_rho, 61, 64), zeg_p = __shfl(pv_rho, 60, 64);
  __syncthreads();
#ifdef RVT_PROF_K4
  const long long tk_b = clock64();  // per-rho tails, burden tails
#endif
  // ---- SKAT-O preparation -------------------------------------------------------------------------------
  double minP = 1.0;
  int minIndex = 0;
  if (skato_quad) {
    skato_select(gs, sh.pvals, &minP, &minIndex);  // (every lane: 11 LDS reads)
    if (lane < kNRho) {
      const double r0 = 1.0 * lane / 10;
      sh.si.rho[lane] = (r0 > 0.999) ? 0.999 : r0;
      sh.si.qminp[lane] = skato_q_by_moment(minP, mo);
      sh.si.tau[lane] = gs.tau[lane];
    }
    if (lane == 11) {
      sh.si.muQ = gs.muQ;
      sh.si.varQ = gs.varQ;
      sh.si.varZeta = gs.varZeta;
      sh.si.df = gs.df;
      sh.si.lambda = lam_zimz;
      sh.si.th = nullptr;
      sh.si.r = n_zimz;
      sh.si.lambda_sum = gs.zimz_lambda_sum;
      sh.si.pre = &sh.pre;
      sh.si.liu = &sh.liu;
      sh.si.lg_half = lgamma(0.5);
    }
    if (lane == 13) sh.liu = liu_prepare(lam_zimz, n_zimz);
  }
#ifdef RVT_PROF_K4
  const long long tk_c = clock64();  // quantiles, Liu moments
#endif
  // The searches of qf() that do not depend on the quantile, once per coefficient set: SKAT-O's on lane 12, SKAT's own on
  // lane 14 — the same code on two lanes, so they run side by side; both fill the memo of their set on the way.
  if (lane == 12 || lane == 14) {
    const int w = (lane == 14) ? 1 : 0;
    const dv_pre_p P = w ? (dv_pre_p)&sh.pre_skat : (dv_pre_p)&sh.pre;
    const dv_memo_p mm = w ? (dv_memo_p)&sh.memo_skat : (dv_memo_p)&sh.memo;
    if (w ? (do_skat && rs[1] > 1) : skato_quad) {
      davies_prelude_t<FAST>(w ? lbs[1] : lbs[0], nullptr, w ? rs[1] : rs[0], 10000, 0.000001, P, mm, w ? lss[1] : lss[0]);
    } else {
      P->valid = false;
      P->fast = FAST;
      P->memo = nullptr;
    }
  }
  __syncthreads();
  const SkatoIntegrand& si = sh.si;
  // One cooperative Davies round evaluates: the quadrature abscissae of this step (lanes < npts), and — in
  // the first round only — SKAT's own Q (lane 62) and the single-variant SKAT-O Q (lane 63).
#ifdef RVT_PROF_K4
  double prof[4] = {0, 0, 0, 0};
  double prof4[4] = {0, 0, 0, 0};   // front, main, QAGS bookkeeping (lane 0)
#else
  double* prof4 = nullptr;
#endif
  auto davies_round = [&](int npts, double a1, double b1, double b2, bool first, int pass, double* extra62,
                          double* extra63) {
    bool active = false;
    int which = 0;
    double c = 0.0, x = 0.0;
    bool skip_zero = false;  // kappa beyond the cut: integrand uses temp = 0
    dv_pre_cp pp = nullptr;
    dv_memo_p mm = nullptr;
    if (lane < npts && skato_quad) {
      x = first ? gk21_abscissa(a1, b2, lane)
                : ((lane < 21) ? gk21_abscissa(a1, b1, lane) : gk21_abscissa(b1, b2, lane - 21));
      if (pass == 0) {
        const double kappa = skato_kappa(si, x);
        if (kappa > si.lambda_sum * 10000) {
          skip_zero = true;
        } else {
          c = (kappa - si.muQ) * sqrt(si.varQ - si.varZeta) / sqrt(si.varQ) + si.muQ;
          active = true;
          pp = (dv_pre_cp)&sh.pre;
          mm = (dv_memo_p)&sh.memo;
        }
      }
    } else if (first && lane == 62 && do_skat) {
      active = true;
      which = 1;
      c = gs.skat_Q;
      pp = (dv_pre_cp)&sh.pre_skat;
      mm = (dv_memo_p)&sh.memo_skat;
    } else if (first && lane == 63 && do_skato && gs.skato_single) {
      active = true;
      which = 0;
      c = gs.Qs[0];
    }
    double nt = 0.0;
    double p = 0.0;
#ifdef RVT_PROF_K4
    const long long tk0 = clock64();
#endif
    if (pass == 0 || first) p = wave_davies_pvalue<FAST>(active, which, lbs, lss, rs, c, pp, mm, lane, L, &nt, prof4);
    terms += nt;
#ifdef RVT_PROF_K4
    const long long tk1 = clock64();
    prof[0] += (double)(tk1 - tk0);
#endif
    if (lane < npts && skato_quad) {
      double val;
      if (pass == 0) {
        double temp = skip_zero ? 0.0 : p;
        if (!skip_zero && (temp <= 0.0 || temp == 1.0)) temp = liu_pvalue_pre(sh.liu, c);
        val = (1.0 - temp) * chisq_density_lg(x, 1.0, si.lg_half);
      } else {
        val = skato_integrand_liu(si, x);
      }
      fv[lane] = val;
    }
    if (first && lane == 62 && do_skat) {
      if (!fam && (p <= 0.0 || p == 1.0)) p = liu_pvalue(lam_skat, n_skat, gs.skat_Q);  // Skat.cpp:100-103
      *extra62 = p;
    }
    if (first && lane == 63 && do_skato && gs.skato_single) *extra63 = p;
#ifdef RVT_PROF_K4
    prof[1] += (double)(clock64() - tk1);
#endif
    __syncthreads();
  };
  double skat_p = 0.0, single_p = 0.0;
  int neval = 0, status = 0, status0 = 0;
#ifdef RVT_PROF_K4
  const long long tk_start = clock64();
#endif
  double integral = 0.0;
  QagsMachine& qm = sh.qm;  // driven by lane 0; the wave-uniform control words sit in sh.ctl
  double* const ctl = sh.ctl;
  for (int pass = 0; pass < 2; ++pass) {
    bool in_lds = true;  // (lane 0) the interval store still fits the LDS arrays
    if (lane == 0) {
      if (skato_quad) {
        QagsWorkspace ws;
        ws.alist = sh.alist;
        ws.blist = sh.blist;
        ws.rlist = sh.rlist;
        ws.elist = sh.elist;
        ws.order = sh.order;
        ws.level = sh.level;
        qm.begin(0., 40., kSkatoEpsAbs, kSkatoEpsRel, kSkatoLimit, ws);
        ctl[3] = qm.running() ? 1.0 : 0.0;
      } else {
        ctl[3] = 0.0;
      }
    }
    __syncthreads();
    bool running = ctl[3] != 0.0;
    if (running || pass == 0) {
      davies_round(running ? 21 : 0, 0., 20., 40., true, pass, &skat_p, &single_p);
      if (running) neval += 21;
      if (lane == 0 && running) {
        qm.first_panel(fv);
        ctl[3] = qm.running() ? 1.0 : 0.0;
        if (qm.running()) qm.bisect(&ctl[0], &ctl[1], &ctl[2]);
      }
      __syncthreads();
      running = running && ctl[3] != 0.0;
    }
    while (running) {
      const double a1 = ctl[0], b1 = ctl[1], b2 = ctl[2];
      __syncthreads();
      davies_round(42, a1, b1, b2, false, pass, &skat_p, &single_p);
      neval += 42;
      if (lane == 0) {
#ifdef RVT_PROF_K4
        const long long tb0 = clock64();
#endif
        if (in_lds && qm.size + 2 >= kQagsLds) {  // the store outgrows LDS: continue in the global workspace
          QagsWorkspace gw = qags_workspace_carve(gd.qags_mem, kSkatoLimit);
          for (int i = 0; i < qm.size; ++i) {
            gw.alist[i] = sh.alist[i];
            gw.blist[i] = sh.blist[i];
            gw.rlist[i] = sh.rlist[i];
            gw.elist[i] = sh.elist[i];
            gw.order[i] = sh.order[i];
            gw.level[i] = sh.level[i];
          }
          gw.order[qm.size] = sh.order[qm.size];  // (sort_after_insert may read one slot past the end)
          qm.w = gw;
          in_lds = false;
        }
        qm.advance(fv, fv + 21);
        ctl[3] = qm.running() ? 1.0 : 0.0;
        if (qm.running()) qm.bisect(&ctl[0], &ctl[1], &ctl[2]);
        RVT_K4_TICK(2, tb0);
      }
      __syncthreads();
      running = ctl[3] != 0.0;
    }
    if (!skato_quad) break;
    __syncthreads();
    if (lane == 0) {
      ctl[0] = qm.result;
      ctl[1] = (double)qm.status;
    }
    __syncthreads();
    integral = ctl[0];
    status = (int)ctl[1];
    __syncthreads();
    if (pass == 0) {
      status0 = status;
      if (status == 0) break;
    } else {
      status0 = status0 * 100 + status;
    }
  }
  skat_p = __shfl(skat_p, 62, 64);
  single_p = __shfl(single_p, 63, 64);
  // total Davies terms over the wave
  for (int off = 32; off > 0; off >>= 1) terms += __shfl_down(terms, off, 64);
  if (lane != 0) return;
  if ((tests & RVT_TEST_CMC) && gs.cmc_ok) out->cmc_p = cmc_p;
  if ((tests & RVT_TEST_ZEGGINI) && gs.zeg_ok) out->zeg_p = zeg_p;
  if (do_skat && fam) {
    out->famskat_ok = 1;
    out->famskat_Q = gs.skat_Q;
    out->famskat_p = skat_p;
    out->skat_nlambda = gs.skat_nlambda;
  } else if (do_skat) {
    out->skat_ok = 1;
    out->skat_Q = gs.skat_Q;
    out->skat_p = skat_p;
  }
  if (do_skato && gs.skato_single) {
    out->skato_ok = 1;
    out->skato_Q = gs.Qs[0];
    out->skato_rho = 0.0;
    out->skato_p = single_p;
  } else if (do_skato) {
    out->skato_qags_status = status0;
    out->skato_qags_neval = neval;
    double rho = (minIndex == 10) ? 0.999 : 1.0 * minIndex / 10;
    if (rho >= 0.999) rho = 1.;
    out->skato_rho = rho;
    out->skato_Q = gs.Qs[minIndex];
    out->skato_p = skato_finish(integral, minP, sh.pvals);
    out->skato_ok = 1;
  }
  out->davies_terms = terms;
#ifdef RVT_PROF_K4
  out->cmc_U = prof[0];                              // cycles inside wave_davies_pvalue
  out->cmc_V = prof[1];                              // cycles in Liu fallback + density + fv store
  out->zeg_U = (double)(clock64() - tk_start);       // cycles from QAGS start to end
  out->zeg_V = (double)neval;
  out->famcmc_U = prof4[0];                          // front
  out->famcmc_V = prof4[1];                          // flattened main integration
  out->famzeg_U = prof4[2];                          // lane-0 QAGS bookkeeping
  out->famzeg_V = (double)(tk_start - tk_entry);     // before the QAGS loop (loads, order, prelude, moments)
  out->famskat_Q = (double)(tk_a - tk_entry);        //   loads, sorted copies, memo clear
  out->famskat_p = (double)(tk_b - tk_a);            //   per-rho tails
  out->famcmc_af = (double)(tk_c - tk_b);            //   min-p, quantiles, Liu moments
  out->famzeg_af = (double)(tk_start - tk_c);        //   the two preludes
  {  // gene_assemble's phases (its own clock readings, left in the gene's statistics)
    const double* at = gd.stats->as_ticks;
    out->vt_minmaf = at[1] - at[0];
    out->vt_maxmaf = at[2] - at[1];
    out->vt_optmaf = at[3] - at[2];
    out->vt_U = at[4] - at[3];
    out->vt_V = at[5] - at[4];
    out->vt_stat = at[6] - at[5];
    out->vt_p = at[7] - at[6];
    out->vt_p_error = at[7] - at[0];
    out->perm_pvalue = at[8] - at[0];  // (thread 0's own share of the partial-statistics loop)
    out->cmc_stat = at[9];             // (wave-parts summed)
  }
#endif
}

}  // namespace rvt
