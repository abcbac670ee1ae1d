// rvtests_amd — per-gene statistics: everything between the MFMA sufficient-statistics pass and the
// p-value kernel.  One workgroup per gene (`Coop`); also compiled for the host test harness.
//
// Input: the gene's sufficient statistics of the UNFLIPPED genotype block G (N x M) against the null
// model,  R = G'·D·[G | X | rr]  (M x (M+d+1)), D = I and rr = res for a quantitative trait,
// D = diag(v) and rr = res/v for a binary one; exact per-column sum / min / max of G; and the
// collapsed-burden partial sums.  From these it reproduces, without touching the N-sized data again:
//   * DataConsolidator::getFlippedToMinorPolymorphicGenotype (src/DataConsolidator.h:128-132;
//     flip rule .cpp:46-69, monomorphic rule .cpp:94-142) — flips applied ALGEBRAICALLY to R
//     (g' = 2 - g), monomorphic columns dropped;
//   * SkatTest weights beta_pdf(maf)^2 and SkatOTest weights beta_pdf(maf) with the reference's
//     column-index quirk (src/Model.h:2644-2661, 2799-2813);
//   * Skat::Fit's Q and the eigenvalues of K_sqrt P0 K_sqrt' in the folded form
//     W½ (G'VG − G'VX (X'VX)^-1 X'VG) W½ (regression/Skat.cpp:47-98);
//   * SkatO::Fit's Q_rho, per-rho eigenvalues of L'(Z1'Z1)L, moments, Z(I−M)Z' eigenvalues, VarZeta,
//     MuQ, VarQ, Df and tau_rho (regression/SkatO.cpp:124-203, 350-418), and FitSKAT for M = 1 (:60-99);
//   * the 1-df score statistics of CMCTest / ZegginiTest (regression/LinearRegressionScoreTest.cpp:209-261,
//     regression/LogisticRegressionScoreTest.cpp:260-300).
#pragma once
#include "rvt_coop.h"
#include "rvt_skato.h"
#include "rvt_types.h"

namespace rvt {

// layout of one burden partial record (per test): U, cVc, count, cVX[0..d-1]
RVT_HD int burden_rec_len(int d) { return 3 + d; }

constexpr int kNEigen = 13;   // eigenproblems per gene: 11 rho, Z(I-M)Z', SKAT
constexpr int kNTridiag = 2;  // dense reductions per gene: the SKAT-O family's shared one, SKAT's own

// per-gene workspace in global memory
struct GeneScratch {
  double* R;     // Mp x Cp   reduced statistics, row-major (ld = Cp)
  double* Wm;    // Mp x Mp   projected matrix S' - T' Cinv T'^T on the kept columns (m x m, column-major)
  double* eig;   // kNTridiag x Mp x Mp   work matrices of the dense reductions (used when they do not fit in LDS)
  double* tri;   // kNTridiag x 2 x Mp    tridiagonal forms (d, e) of the SKAT-O family and of SKAT
  double* vecs;  // 16 * Mp doubles of vector scratch (assemble stage)
  double* bw;    // 2 * Mp: sqrt(SKAT weight), SKAT-O weight (filtered index)
  double* rowsum;  // Mp: row sums of A = B Wm B / 2
  int* ivec;     // 2 * Mp ints
};
RVT_HD size_t gene_scratch_doubles(int Mp, int Cp) {
  return (size_t)Mp * Cp + (size_t)(1 + kNTridiag) * Mp * Mp + 16 * (size_t)Mp + 3 * (size_t)Mp + (size_t)Mp +
         (size_t)kNTridiag * 2 * Mp;
}
RVT_HD GeneScratch gene_scratch_carve(double* mem, int Mp, int Cp) {
  GeneScratch s;
  s.R = mem;
  s.Wm = s.R + (size_t)Mp * Cp;
  s.eig = s.Wm + (size_t)Mp * Mp;
  s.vecs = s.eig + (size_t)kNTridiag * Mp * Mp;
  s.bw = s.vecs + 16 * (size_t)Mp;
  s.rowsum = s.bw + 2 * (size_t)Mp;
  s.ivec = (int*)(s.rowsum + (size_t)Mp);
  s.tri = (double*)(s.ivec) + (size_t)Mp;  // ivec holds 2*Mp ints = Mp doubles
  return s;
}

// SkatOImpl::getEigen filter on ascending eigenvalues `ev` (regression/SkatO.cpp:350-382).
// Writes kept values in DEcreasing order to `out`; returns count, or -1 when none is positive.
RVT_HD int skato_filter_eigen(const double* ev, int n, double* out) {
  int numNonZero = 0;
  double sumNonZero = 0.;
  for (int i = 0; i < n; ++i)
    if (ev[i] > 0) {
      ++numNonZero;
      sumNonZero += ev[i];
    }
  if (numNonZero == 0) return -1;
  const double t = sumNonZero / numNonZero / 100000;
  int numKeep = n;
  for (int i = 0; i < n; ++i) {
    if (ev[i] < t)
      --numKeep;
    else
      break;
  }
  for (int i = 0; i < numKeep; ++i) out[i] = ev[n - 1 - i];
  return numKeep;
}

RVT_HD double skato_rho_value(int i) {
  const double r0 = 1.0 * i / 10;
  return (r0 > 0.999) ? 0.999 : r0;  // capRhos, SkatO.cpp:436-446
}

// ======================================================================================================
// Stage A (one workgroup per gene): reduce the partial statistics, flags, flip algebra, projection,
// weights, Q statistics, tau, burden score statistics.  Leaves Wm, the weights and the row sums of
// A = B Wm B / 2 in the gene's scratch for stage B.
//   parts:   P partial matrices of Mp x Cp doubles (row-major); only tiles with tile_col >= tile_row hold data
//   colstat: P x 3 x Mp  (sum, min, max) partials
//   bparts:  PB x 2 x burden_rec_len(d) partial burden sums (CMC, Zeggini) — may be null when no burden test
// Hard-call path with masked entries (suffstat_hc.hip.h; mean-imputed columns G_j = H_j + mu_j m_j): `hcm` carries the
// per-wave-part images of the 16-bit counters of P' = (H + 4m)'m and Q = m'm and the wave-part flags; colstat then has
// kHcRows rows per part (sum of H, min / max over the hard calls, masked count, OR / AND of the masked bit patterns) and
// the G'DG block of `parts` holds C = (H + 4m)'(H + 4m).  Recovered here, in exact integer arithmetic up to one rounding
// per product:  P = P' - 4Q,  H'H = C - 4(P + P') - 16 Q,  G'G = H'H + P diag(mu) + diag(mu) P' + diag(mu) Q diag(mu).
// extra_status: OR-ed into the gene's status word.
// ======================================================================================================
constexpr unsigned kStatusHandedBack = 0x100u;  // internal bookkeeping (cleared before a record is returned): the gene
                                                // started on the hard-call kernel and was computed by the fp64 kernel

RVT_HD double rvt_bits_to_double(unsigned long long b) {
  double x;
  __builtin_memcpy(&x, &b, sizeof(x));
  return x;
}

struct HcMasked {
  const unsigned* pq;      // P x pq_words packed counters ([tile][half][lane], tiles: P' MT x MT, then Q upper triangle)
  const unsigned* wflags;  // P flags: bit 0 = the part's image was written
  int pq_words;
  double lat_den;  // > 0: lattice dosages (suffstat_lat.hip.h) — the G'G block and the column sums are integers, scaled here
  // weighted hard-call path (suffstat_hcx.hip.h): the gene's masked-entry tables as 64-bit integers — P = m'VH (Mp x Mp, row =
  // the masked column) and Q = m'Vm (Mp x Mp, upper triangle) in units of 2^-42, R = m'V[X | res] (Mp x 16) in units of
  // xscale[k]; `parts` holds H'VH and H'V[X | res] (a masked entry counted as 0)
  const unsigned long long* pqw = nullptr;
  const double* xscale = nullptr;
};
constexpr int kHcRows = 6;  // == kHcColstatRows (suffstat_hc.hip.h)

#if defined(RVT_PROF_K4) && defined(__HIP_DEVICE_COMPILE__)
#define RVT_AS_TICK(k) do { if (co.tid == 0) out->as_ticks[k] = (double)clock64(); } while (0)
#else
#define RVT_AS_TICK(k) do { } while (0)
#endif
RVT_HD void gene_assemble(const Coop& co, const NullConsts& nc, int M, int Mp, int Cp, const double* parts, int P,
                          const double* colstat, const double* bparts, int PB, const double* af,
                          const rvt_params& prm, unsigned tests, GeneScratch ws, GeneStats* out, int* flip_out,
                          int* kept_out, const HcMasked* hcm = nullptr, unsigned extra_status = 0u, bool parts_reduced = false) {
  const int d = nc.d;
  const int ldr = Cp;
  double* R = ws.R;
  RVT_AS_TICK(0);
  // (parts_reduced: gene_reduce_parts_kernel has already put the sums of step 1 into R — the same additions in the same order,
  //  spread over several workgroups per gene)
  if (!parts_reduced)
  // ---- 1. reduce the partial statistics (fixed order => deterministic) ----------------------------
  // (eight entries per thread and pass: their 8 P loads are independent and in flight together — with one entry per pass a
  //  thread waited out P round trips to HBM per entry; the order of the sum over p is unchanged: bit-reproducible)
  {
    const int total = Mp * Cp;
    const size_t stride = (size_t)Mp * Cp;
    const double lat2 = (hcm && hcm->lat_den > 0.0) ? hcm->lat_den * hcm->lat_den : 0.0;
    for (int idx0 = co.tid; idx0 < total; idx0 += 8 * co.nt) {
      double s8[8];
      bool use[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = idx0 + u * co.nt;
        s8[u] = 0.0;
        use[u] = idx < total && ((idx % Cp) >> 4) >= ((idx / Cp) >> 4);
      }
      for (int p0 = 0; p0 < P; p0 += 4) {  // four wave-parts x eight entries: 32 loads issued before the first sum
        double t[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double* src = parts + (size_t)(p0 + q) * stride;
#pragma unroll
          for (int u = 0; u < 8; ++u) t[q][u] = (use[u] && p0 + q < P) ? src[idx0 + u * co.nt] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (p0 + q < P) s8[u] += t[q][u];  // (p ascending, as always)
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = idx0 + u * co.nt;
        if (idx >= total) continue;
        double sv = s8[u];
        // lattice dosages: the sum of the integer tiles K'K is exact; G'G = K'K / den^2 with one rounding
        if (use[u] && lat2 > 0.0 && (idx % Cp) < M) sv /= lat2;
        R[idx] = sv;
      }
    }
  }
  RVT_AS_TICK(8);
#if defined(RVT_PROF_K4) && defined(__HIP_DEVICE_COMPILE__)
  if (co.tid == 0) out->as_ticks[9] = (double)P;
#endif
  double* colsum = ws.vecs;         // [Mp]
  double* cmin = ws.vecs + Mp;      // [Mp]
  double* cmax = ws.vecs + 2 * Mp;  // [Mp]
  double* sgn = ws.vecs + 3 * Mp;   // [Mp]  +1 / -1
  double* shf = ws.vecs + 4 * Mp;   // [Mp]   0 / 2
  double* ut = ws.vecs + 5 * Mp;    // weighted scores
  double* bw_skat = ws.bw;          // sqrt(beta_pdf^2)
  double* bw_skato = ws.bw + Mp;    // beta_pdf
  double* rowsum = ws.rowsum;
  int* kidx = ws.ivec;              // kept column list
  double* muv = ws.vecs + 6 * Mp;   // [Mp]  imputed value of a column with masked entries (hard-call path)
  double* cmv = ws.vecs + 7 * Mp;   // [Mp]  number of masked entries
  const int cs_rows = hcm ? kHcRows : 3;
  for (int j = co.tid; j < M; j += co.nt) {
    double s = 0.0, mn = INFINITY, mx = -INFINITY, cm = 0.0;
    unsigned long long orb = 0ull;
    for (int p = 0; p < P; ++p) {
      const double* c = colstat + (size_t)p * cs_rows * Mp;
      s += c[j];
      mn = fmin(mn, c[Mp + j]);
      mx = fmax(mx, c[2 * Mp + j]);
      if (hcm) {
        cm += c[3 * Mp + j];
        orb |= reinterpret_cast<const unsigned long long*>(c)[4 * Mp + j];
      }
    }
    if (hcm) {
      double mu = 0.0;
      if (cm > 0.0) {  // (gene_flags_hc_kernel has checked that every masked entry of the column holds these bits)
        mu = rvt_bits_to_double(orb);
        s += cm * mu;
        mn = fmin(mn, mu);
        mx = fmax(mx, mu);
      }
      muv[j] = mu;
      cmv[j] = cm;
      if (hcm->lat_den > 0.0) s /= hcm->lat_den;  // (the exact integer sum of K)
    }
    colsum[j] = s;
    cmin[j] = mn;
    cmax[j] = mx;
    const bool flip = !(s <= (double)nc.N);  // convertToMinorAlleleCount: keep when s <= rows
    sgn[j] = flip ? -1.0 : 1.0;
    shf[j] = flip ? 2.0 : 0.0;
    if (flip_out) flip_out[j] = flip ? 1 : 0;
  }
  co.sync();
  RVT_AS_TICK(1);
  // hard-call path: masked-entry corrections of the G'G block (upper tiles; the completion below mirrors them)
  if (hcm && hcm->pq) {
    bool any = false;
    for (int p = 0; p < P; ++p) any = any || (hcm->wflags[p] & 1u);
    if (any) {
      const int MT = Mp >> 4, ntiles = MT * MT + MT * (MT + 1) / 2;
      double* Pp = ws.eig;                    // Mp x Mp: P'
      double* Qq = ws.eig + (size_t)Mp * Mp;  // Mp x Mp: Q (both triangles)
      for (int idx = co.tid; idx < ntiles * 256; idx += co.nt) {
        const int tile = idx >> 8, e = idx & 255, ri = e >> 4, ci = e & 15;
        const int lane = 16 * (ri >> 2) + ci, reg = ri & 3;
        const size_t word = (size_t)(tile * 2 + (reg >> 1)) * 64 + lane;
        const int sh = 16 * (reg & 1);
        long long acc = 0;
        for (int p = 0; p < P; ++p)
          if (hcm->wflags[p] & 1u) acc += (long long)((hcm->pq[(size_t)p * hcm->pq_words + word] >> sh) & 0xffffu);
        if (tile < MT * MT) {
          const int r = tile / MT, c = tile % MT;
          Pp[(size_t)(r * 16 + ri) * Mp + c * 16 + ci] = (double)acc;
        } else {
          int t = tile - MT * MT, r = 0;
          while (t >= MT - r) {
            t -= MT - r;
            ++r;
          }
          const int c = r + t;
          Qq[(size_t)(r * 16 + ri) * Mp + c * 16 + ci] = (double)acc;
          if (c != r) Qq[(size_t)(c * 16 + ci) * Mp + r * 16 + ri] = (double)acc;
        }
      }
      co.sync();
      for (int idx = co.tid; idx < M * M; idx += co.nt) {
        const int i = idx / M, j = idx % M;
        if ((j >> 4) < (i >> 4)) continue;
        const double q = Qq[(size_t)i * Mp + j];
        const double pij = Pp[(size_t)i * Mp + j] - 4.0 * q, pji = Pp[(size_t)j * Mp + i] - 4.0 * q;
        const double hh = R[(size_t)i * ldr + j] - 4.0 * (pij + pji) - 16.0 * q;  // exact: integers below 2^53
        R[(size_t)i * ldr + j] = hh + muv[j] * pij + muv[i] * pji + (muv[i] * muv[j]) * q;
      }
      co.sync();
    }
  }
  // weighted hard-call path: G'VG = H'VH + P diag(mu) + diag(mu) P' + diag(mu) Q diag(mu), G'V[X | res] = H'V[X | res] + diag(mu) R
  // (exact integers, one rounding per product)
  if (hcm && hcm->pqw) {
    bool any = false;
    for (int p = 0; p < P; ++p) any = any || (hcm->wflags[p] & 1u);
    if (any) {
      const long long* Pw = reinterpret_cast<const long long*>(hcm->pqw);
      const long long* Qw = Pw + (size_t)Mp * Mp;
      const long long* Rw = Qw + (size_t)Mp * Mp;
      for (int idx = co.tid; idx < M * M; idx += co.nt) {
        const int i = idx / M, j = idx % M;
        if ((j >> 4) < (i >> 4)) continue;
        if (muv[i] == 0.0 && muv[j] == 0.0) continue;
        const double pij = (double)Pw[(size_t)i * Mp + j] * 0x1p-42;  // m_i'V H_j
        const double pji = (double)Pw[(size_t)j * Mp + i] * 0x1p-42;  // m_j'V H_i
        const double q = (double)Qw[(size_t)(i < j ? i : j) * Mp + (i < j ? j : i)] * 0x1p-42;
        R[(size_t)i * ldr + j] += muv[j] * pji + muv[i] * pij + (muv[i] * muv[j]) * q;
      }
      for (int idx = co.tid; idx < M * (d + 1); idx += co.nt) {
        const int i = idx / (d + 1), k = idx % (d + 1);
        if (muv[i] != 0.0) R[(size_t)i * ldr + M + k] += muv[i] * ((double)Rw[(size_t)i * 16 + k] * hcm->xscale[k]);
      }
      co.sync();
    }
  }
  // symmetric completion of the G'DG block
  for (int idx = co.tid; idx < M * M; idx += co.nt) {
    const int i = idx / M, j = idx % M;
    if (j < i) R[(size_t)i * ldr + j] = R[(size_t)j * ldr + i];  // lower triangle := upper (exact symmetry)
  }
  co.sync();
  RVT_AS_TICK(2);
  // ---- 2. polymorphic columns ----------------------------------------------------------------------
  if (co.tid == 0) {
    int m = 0, nf = 0;
    for (int j = 0; j < M; ++j) {
      const bool mono = (cmin[j] == cmax[j]);
      if (kept_out) kept_out[j] = mono ? 0 : 1;
      if (!mono) kidx[m++] = j;
      if (shf[j] != 0.0) ++nf;
    }
    kidx[Mp] = m;  // stash
    out->n_variants = M;
    out->n_poly = m;
    out->flip_count = nf;
    out->status = ((m == 0) ? RVT_ST_NO_POLY : 0) | extra_status;
    out->skato_ok = 0;
    out->skato_single = 0;
    out->skat_nlambda = 0;
    out->zimz_nlambda = 0;
    out->skat_lambda_off = 0;
    out->zimz_lambda_off = M;
    out->cmc_ok = out->zeg_ok = 0;
    out->cmc_nonref = 0;
    // (the numbers of a test that is not run — no polymorphic variant, a test not asked for — are zeros in the record, not what
    //  the work space held: records are reproducible bit for bit whatever ran before)
    out->skat_Q = 0.0;
    out->cmc_U = out->cmc_V = out->cmc_stat = 0.0;
    out->zeg_U = out->zeg_V = out->zeg_stat = 0.0;
    for (int k = 0; k < kNEigen; ++k) out->eig_ok[k] = 0;
  }
  co.sync();
  const int m = kidx[Mp];
  // ---- burden score statistics (independent of the rest) ---------------------------------------------
  if (co.tid == 0 && bparts && m > 0) {
    const int rl = burden_rec_len(d);
    for (int t = 0; t < 2; ++t) {
      if (!(tests & (t == 0 ? RVT_TEST_CMC : RVT_TEST_ZEGGINI))) continue;
      double U = 0, cvc = 0, cnt = 0, cz[RVT_MAX_COV];
      for (int k = 0; k < d; ++k) cz[k] = 0;
      for (int p = 0; p < PB; ++p) {
        const double* b = bparts + ((size_t)p * 2 + t) * rl;
        U += b[0];
        cvc += b[1];
        cnt += b[2];
        for (int k = 0; k < d; ++k) cz[k] += b[3 + k];
      }
      double q = 0;
      for (int k = 0; k < d; ++k) {
        double s = 0;
        for (int l = 0; l < d; ++l) s += nc.Cinv[k * d + l] * cz[l];
        q += cz[k] * s;
      }
      const double SS = cvc - q;
      double V, stat;
      int ok = 1;
      if (!nc.binary) {
        V = SS * nc.sigma2;
        double SSi = 1.0 / SS;
        SSi /= nc.sigma2;
        stat = U * SSi * U;
      } else {
        V = SS;
        stat = U * (1.0 / SS) * U;
      }
      if (!(SS > 0) || stat < 0) ok = 0;
      if (t == 0) {
        out->cmc_U = U;
        out->cmc_V = V;
        out->cmc_stat = stat;
        out->cmc_ok = ok;
        out->cmc_nonref = (int)cnt;
        if (!ok) out->status |= RVT_ST_CMC_FAIL;
      } else {
        out->zeg_U = U;
        out->zeg_V = V;
        out->zeg_stat = stat;
        out->zeg_ok = ok;
        if (!ok) out->status |= RVT_ST_ZEG_FAIL;
      }
    }
  }
  if (m == 0 || !(tests & (RVT_TEST_SKAT | RVT_TEST_SKATO | RVT_TEST_ANALYTICVT))) {
    co.sync();
    return;
  }
  RVT_AS_TICK(3);
  // ---- 3. flip algebra:  g' = sgn*g + shf*1 ----------------------------------------------------------
  //   S'_ij = s_i s_j S_ij + s_i t_j g1_i + t_i s_j g1_j + t_i t_j c00,  g1 = G'D1 = T[:,0]
  //   T'_ik = s_i T_ik + t_i C[0][k],   u'_i = s_i u_i + t_i * sum(res)
  const double c00 = nc.C[0];
  for (int idx = co.tid; idx < M * M; idx += co.nt) {
    const int i = idx / M, j = idx % M;
    if (shf[i] != 0.0 || shf[j] != 0.0) {
      const double g1i = R[(size_t)i * ldr + M], g1j = R[(size_t)j * ldr + M];
      R[(size_t)i * ldr + j] =
          sgn[i] * sgn[j] * R[(size_t)i * ldr + j] + sgn[i] * shf[j] * g1i + shf[i] * sgn[j] * g1j + shf[i] * shf[j] * c00;
    }
  }
  co.sync();
  for (int i = co.tid; i < M; i += co.nt) {
    if (shf[i] != 0.0) {
      for (int k = 0; k < d; ++k) R[(size_t)i * ldr + M + k] = sgn[i] * R[(size_t)i * ldr + M + k] + shf[i] * nc.C[k];
      R[(size_t)i * ldr + M + d] = sgn[i] * R[(size_t)i * ldr + M + d] + shf[i] * nc.rsum;
    }
  }
  co.sync();
  RVT_AS_TICK(4);
  // ---- 4. projected matrix  Wm = S' − T' Cinv T'ᵀ  on the kept columns (column-major m x m) ----------
  double* Wm = ws.Wm;
  for (int idx = co.tid; idx < m * m; idx += co.nt) {
    const int a = idx % m, b = idx / m;
    const int ia = kidx[a], ib = kidx[b];
    double q = 0.0;
    for (int k = 0; k < d; ++k) {
      double s = 0.0;
      for (int l = 0; l < d; ++l) s += nc.Cinv[k * d + l] * R[(size_t)ib * ldr + M + l];
      q += R[(size_t)ia * ldr + M + k] * s;
    }
    Wm[(size_t)b * m + a] = R[(size_t)ia * ldr + ib] - q;
  }
  // ---- 5. weights (quirk: filtered position a reads the counter of unfiltered column a) ----------------
  for (int a = co.tid; a < m; a += co.nt) {
    double freq = af[a];
    if (freq > 0.5) freq = 1.0 - freq;
    double w1 = 0.0, w2 = 0.0;
    if (freq > 1e-30) {
      w1 = beta_density(freq, prm.skat_beta1, prm.skat_beta2);
      w1 *= w1;
      w2 = beta_density(freq, prm.skato_beta1, prm.skato_beta2);
    }
    bw_skat[a] = sqrt(w1);
    bw_skato[a] = w2;
  }
  co.sync();
  RVT_AS_TICK(5);
  // ---- 6. SKAT Q = || W½ G'ᵀ r ||²      (Skat.cpp:52) ----------------------------------------------------
  if ((tests & RVT_TEST_SKAT) && co.tid == 0) {
    double Q = 0.0;
    for (int a = 0; a < m; ++a) {
      const double s = bw_skat[a] * R[(size_t)kidx[a] * ldr + M + d];
      Q += s * s;
    }
    out->skat_Q = Q;
  }
  if (!(tests & RVT_TEST_SKATO)) {
    co.sync();
    return;
  }
  // ---- 7. SKAT-O scalars: Q_rho, row sums of A = B Wm B / 2, tau_rho -------------------------------------
  for (int a = co.tid; a < m; a += co.nt) {
    ut[a] = bw_skato[a] * R[(size_t)kidx[a] * ldr + M + d];
    double s = 0.0;
    for (int q = m - 1; q >= 0; --q) s += bw_skato[a] * Wm[(size_t)q * m + a] * bw_skato[q] / 2.0;
    rowsum[a] = s;
  }
  co.sync();
  RVT_AS_TICK(6);
  if (co.tid == 0) {
    double s2;
    if (nc.binary)
      s2 = 1.0;
    else {
      s2 = sqrt(nc.rss);
      s2 = (s2 * s2) / (double)(nc.N - 1);
    }
    double su = 0.0, su2 = 0.0;
    for (int a = 0; a < m; ++a) {
      su += ut[a];
      su2 += ut[a] * ut[a];
    }
    if (m == 1) {
      // FitSKAT (SkatO.cpp:60-99): Q = u²/s2/2, W = A (1 x 1), Davies on its single eigenvalue (=> Liu)
      out->skato_single = 1;
      double Q = ut[0] * ut[0];
      if (!nc.binary) Q /= nc.rss / (double)(nc.N - 1);  // squaredNorm()/(nPeople-1)
      Q /= 2.;
      out->Qs[0] = Q;
      out->skato_ok = 1;  // stage B (problem 11) clears it when the eigenvalue is not positive
    } else {
      double tot = 0.0, r2 = 0.0;
      for (int a = 0; a < m; ++a) {
        tot += rowsum[a];
        r2 += rowsum[a] * rowsum[a];
      }
      // tau_rho = m² rho z_norm + (1-rho) ||z̄'Z1||² / z_norm, z_norm = 1'A1/m², z̄'Z1 = (A1)'/m   (SkatO.cpp:198-203)
      const double z_norm = tot / ((double)m * (double)m);
      const double zz = r2 / ((double)m * (double)m);
      for (int i = 0; i < kNRho; ++i) {
        const double rh = skato_rho_value(i);
        double q = (1.0 - rh) * su2 + rh * (su * su);
        q /= s2;
        q /= 2.0;
        out->Qs[i] = q;
        out->tau[i] = (double)(m * m) * rh * z_norm + (1.0 - rh) * zz / z_norm;
      }
      out->skato_ok = 1;  // stage B clears eig_ok[k] when a getEigen finds no positive eigenvalue
    }
  }
  co.sync();
  RVT_AS_TICK(7);
}

// ======================================================================================================
// Stage B: eigenvalues.  The reference solves 13 dense symmetric eigenproblems per gene
//   k = 0..10   L'(Z1'Z1/2)L for rho_k, L L' = R_rho = (1-rho) I + rho 11'   (SkatO.cpp:163-175)
//   k = 11      Z(I-M)Z' = A - (A1)(A1)'/(1'A1)                               (SkatO.cpp:178-195)
//   k = 12      K_sqrt P0 K_sqrt'                                             (Skat.cpp:47-98)
// with A = Z1'Z1/2 = B Wm B / 2.  Twelve of them are rank-one modifications of the SAME matrix, and they
// share one tridiagonal form:
//   * L'AL is similar to R^1/2 A R^1/2 and R^1/2 = sqrt(1-rho) (I + (kappa-1) e e'), e = 1/sqrt(m),
//     kappa^2 = (1-rho+m rho)/(1-rho).  With the Householder reflector H that maps e onto the first unit
//     vector, H R^1/2 H = sqrt(1-rho) D, D = diag(kappa, 1, .., 1), so L'AL ~ (1-rho) D (HAH) D.
//   * Householder tridiagonalisation T = Q'(HAH)Q never touches the first coordinate (Q = diag(1, Q')),
//     so Q commutes with D and (1-rho) D T D — T with t11 scaled by (1-rho+m rho), t12 by
//     sqrt((1-rho)(1-rho+m rho)) and the rest by (1-rho) — is the tridiagonal form for EVERY rho.
//   * H (A1) = -sqrt(m) (HAH) e1, so in the same basis Z(I-M)Z' becomes T - (T e1)(T e1)'/t11: row and
//     column 1 vanish (the exact zero eigenvalue, which the reference's filter drops) and the rest is the
//     trailing tridiagonal of T with its first diagonal entry replaced by t22 - t12^2/t11.
// The SKAT matrix equals 2 v A when SKAT and SKAT-O use the same weights (the defaults) and then reuses T as
// well; otherwise it gets its own reduction.  So: stage B1 = one (or two) dense Householder reductions per
// gene, stage B2 = 13 tridiagonal bisections (one workgroup each), instead of 13 dense reductions.
// All of this is orthogonal similarity: the eigenvalues are those of the reference's matrices to rounding.
// ======================================================================================================

// B <- H B H for the reflector H with H 1 = -sqrt(n) e1 (B symmetric n x n, column-major, full storage)
RVT_HD void coop_reflect_ones(const Coop& co, double* B, int n, double* v, double* w) {
  const double nrm = sqrt((double)n);
  const double tau = (nrm + 1.0) / nrm;  // alpha = 1, beta = -sqrt(n): tau = (beta - alpha) / beta
  const double scal = 1.0 / (1.0 + nrm);
  for (int i = co.tid; i < n; i += co.nt) v[i] = (i == 0) ? 1.0 : scal;
  co.sync();
  double dotpart = 0.0;
  for (int i = co.tid; i < n; i += co.nt) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += B[(size_t)j * n + i] * v[j];
    s *= tau;
    w[i] = s;
    dotpart += s * v[i];
  }
  const double pv = co.sum(dotpart);
  const double a2 = -0.5 * tau * pv;
  for (int i = co.tid; i < n; i += co.nt) w[i] = w[i] + a2 * v[i];
  co.sync();
  for (int idx = co.tid; idx < n * n; idx += co.nt) {
    const int i = idx % n, j = idx / n;
    B[idx] -= v[i] * w[j] + w[i] * v[j];
  }
  co.sync();
}

// do SKAT and SKAT-O weigh the variants identically?  (every thread gets the same answer)
RVT_HD bool skat_shares_weights(const GeneScratch& ws, int Mp, int m) {
  for (int i = 0; i < m; ++i)
    if (ws.bw[i] != ws.bw[Mp + i]) return false;
  return true;
}

// Stage B1.  which = 0: SKAT-O family (also VarZeta); which = 1: SKAT (skipped when it can share).
// `Bm` is the m x m work matrix (LDS when it fits, the gene's scratch otherwise); `vec` >= 4*m doubles.
RVT_HD void gene_tridiag(const Coop& co, const NullConsts& nc, int which, int M, int Mp, unsigned tests,
                         GeneScratch ws, double* Bm, double* vec, GeneStats* out) {
  const int m = out->n_poly;
  if (m < 2) return;  // single-variant genes need no reduction
  const double* Wm = ws.Wm;
  const double* bw_skat = ws.bw;
  const double* bw_skato = ws.bw + Mp;
  double* td = vec;          // tridiagonal d
  double* te = vec + m;      // tridiagonal e
  double* hv = vec + 2 * m;  // householder v
  double* hw = vec + 3 * m;  // householder w
  double* tri = ws.tri + (size_t)which * 2 * Mp;
  if (which == 0) {
    if (!(tests & RVT_TEST_SKATO) && !((tests & RVT_TEST_SKAT) && skat_shares_weights(ws, Mp, m))) return;
    const double* rowsum = ws.rowsum;
    double tot = 0.0;
    for (int a = 0; a < m; ++a) tot += rowsum[a];
    double vzpart = 0.0;
    for (int idx = co.tid; idx < m * m; idx += co.nt) {
      const int i = idx % m, j = idx / m;
      const double aij = bw_skato[i] * Wm[idx] * bw_skato[j] / 2.0;
      const double zmz = rowsum[i] * rowsum[j] / tot;
      Bm[idx] = aij;
      vzpart += zmz * (aij - zmz);
    }
    const double vz = co.sum(vzpart);
    if (co.tid == 0) out->varZeta = 4.0 * vz;
    co.sync();
    coop_reflect_ones(co, Bm, m, hv, hw);
  } else {
    if (!(tests & RVT_TEST_SKAT) || skat_shares_weights(ws, Mp, m)) return;
    const double vscale = nc.binary ? 1.0 : nc.sigma2;  // quantitative: statistics were unweighted, v = sigma2
    for (int idx = co.tid; idx < m * m; idx += co.nt) {
      const int a = idx % m, b = idx / m;
      Bm[idx] = bw_skat[a] * (vscale * Wm[idx]) * bw_skat[b];
    }
    co.sync();
  }
  coop_tridiagonalize(co, Bm, m, td, te, hv, hw);
  for (int i = co.tid; i < m; i += co.nt) {
    tri[i] = td[i];
    tri[Mp + i] = (i < m - 1) ? te[i] : 0.0;
  }
  co.sync();
}

// Moments of a rho problem WITHOUT its eigenvalues (round 5).  The reference uses the eigenvalues of L'AL only through their
// power sums (getMoment, SkatO.cpp:383-418) after dropping those below t = mean(positive) / 1e5 (getEigen, :350-382) — and the
// power sums of ALL eigenvalues are traces of powers of the tridiagonal form, O(m) sums of its entries.  They are the
// reference's sums whenever what the filter drops is rounding, which three Sturm counts certify: no eigenvalue in
// [x_lo, x_hi), x_lo = 1e-13 tr, x_hi = tr / 1e5 (t lies in (x_lo, x_hi] whatever sign the zero-level eigenvalues take: at
// least one of the m eigenvalues counts as positive, at most all), none below -x_lo.  Then the dropped ones change a sum by
// less than m 1e-13 of it.  Otherwise (false) the caller computes the eigenvalues and filters them as the reference does.
// d, e2: the SCALED tridiagonal (by 2^-sh; couplings squared and floored) of coop_tridiag_scale / gene_spectrum_all's stage C.
RVT_HD bool skato_moment_by_trace(const double* d, const double* e2, int n, int sh, SkatoMoment* mo) {
  double p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0;
  for (int i = 0; i < n; ++i) {
    const double di = d[i], d2 = di * di;
    p1 += di;
    p2 += d2;
    p3 += d2 * di;
    p4 += d2 * d2;
    if (i < n - 1) {
      const double q = e2[i], dn = d[i + 1];
      p2 += 2.0 * q;
      p3 += 3.0 * q * (di + dn);
      p4 += 4.0 * q * (d2 + di * dn + dn * dn) + 2.0 * q * q;
      if (i < n - 2) p4 += 4.0 * q * e2[i + 1];
    }
  }
  if (!(p1 > 0.0) || !(p2 > 0.0)) return false;
  const double x_hi = p1 * 1e-5, x_lo = p1 * 1e-13;
  const int c_hi = sturm_count(d, e2, n, x_hi), c_lo = sturm_count(d, e2, n, x_lo), c_neg = sturm_count(d, e2, n, -x_lo);
  if (c_hi != c_lo || c_neg != 0 || c_hi >= n) return false;
  *mo = skato_moment_from_sums(ldexp(p1, sh), ldexp(p2, 2 * sh), ldexp(p3, 3 * sh), ldexp(p4, 4 * sh));
  return true;
}

// Stage B2 (one workgroup per gene AND eigenproblem k): the k-th tridiagonal from stage B1's, bisection for all
// eigenvalues, the reference's eigenvalue filter and moments.  `vec` >= 4*m doubles.
RVT_HD void gene_spectrum(const Coop& co, const NullConsts& nc, int k, int M, int Mp, unsigned tests, GeneScratch ws,
                          double* vec, GeneStats* out, double* lambda_out) {
  const int m = out->n_poly;
  if (m == 0) return;
  const bool is_skat = (k == 12);
  if (is_skat && !(tests & RVT_TEST_SKAT)) return;
  if (!is_skat && !(tests & RVT_TEST_SKATO)) return;
  if (!is_skat && m == 1 && k != 11) return;  // single-variant shortcut only needs the one "eigenvalue"
  const double* Wm = ws.Wm;
  const double* bw_skat = ws.bw;
  const double* bw_skato = ws.bw + Mp;
  double* ev = vec;          // eigenvalues ascending
  double* td = vec + m;      // this problem's tridiagonal d
  double* te = vec + 2 * m;  // ... and e
  double* tmpv = vec + 3 * m;
  int n = m;
  if (m == 1) {
    if (co.tid == 0) {
      if (is_skat) {
        const double vscale = nc.binary ? 1.0 : nc.sigma2;
        ev[0] = bw_skat[0] * (vscale * Wm[0]) * bw_skat[0];
      } else {
        const double lam = bw_skato[0] * Wm[0] * bw_skato[0] / 2.0;
        if (lam > 0) {
          lambda_out[M] = lam;
          out->zimz_nlambda = 1;
          out->zimz_lambda_sum = lam;
          out->eig_ok[11] = 1;
        } else {
          out->eig_ok[11] = 0;
        }
      }
    }
    co.sync();
    if (!is_skat) return;
  } else {
    const double* d0 = ws.tri;
    const double* e0 = ws.tri + Mp;
    if (is_skat) {
      if (skat_shares_weights(ws, Mp, m)) {
        const double sc = 2.0 * (nc.binary ? 1.0 : nc.sigma2);
        for (int i = co.tid; i < m; i += co.nt) {
          td[i] = sc * d0[i];
          te[i] = sc * e0[i];
        }
      } else {
        for (int i = co.tid; i < m; i += co.nt) {
          td[i] = ws.tri[2 * Mp + i];
          te[i] = ws.tri[3 * Mp + i];
        }
      }
    } else if (k == 11) {
      n = m - 1;
      if (!(d0[0] > 0.0)) {  // 1'A1 <= 0: the reference's M is 0/0 and every eigenvalue NaN
        if (co.tid == 0) out->eig_ok[11] = 0;
        co.sync();
        return;
      }
      for (int i = co.tid; i < n; i += co.nt) {
        td[i] = (i == 0) ? d0[1] - e0[0] * e0[0] / d0[0] : d0[i + 1];
        te[i] = e0[i + 1];
      }
    } else {
      const double rh = skato_rho_value(k);
      const double s1 = 1.0 - rh, sm = 1.0 - rh + (double)m * rh;
      const double s1m = sqrt(s1 * sm);
      for (int i = co.tid; i < m; i += co.nt) {
        td[i] = (i == 0) ? sm * d0[0] : s1 * d0[i];
        te[i] = (i == 0) ? s1m * e0[0] : s1 * e0[i];
      }
    }
    co.sync();
    const TridiagScale ts = coop_tridiag_scale(co, td, te, n);
    if (!is_skat && k != 11) {  // a rho problem: its moments from the traces when the filter provably drops rounding only
      SkatoMoment mo;
      if (skato_moment_by_trace(td, te, n, ts.sh, &mo)) {  // (every thread computes the same answer)
        if (co.tid == 0) {
          out->mom_mu[k] = mo.muQ;
          out->mom_var[k] = mo.varQ;
          out->mom_df[k] = mo.df;
          out->eig_ok[k] = 1;
        }
        co.sync();
        return;
      }
    }
    coop_tridiag_eigvals_scaled(co, td, te, n, ts, ev);
  }
  if (co.tid == 0) {
    if (is_skat) {
      double* lam_skat = lambda_out;
      const int r_ub = (nc.N < (int64_t)m) ? (int)nc.N : m;
      int r = 0;
      for (int i = n - 1; i >= 0; --i) {
        if (ev[i] > 1e-30 && r < r_ub) {
          lam_skat[r++] = ev[i];
        } else
          break;
      }
      out->skat_nlambda = r;
      out->eig_ok[12] = 1;
    } else if (k == 11) {
      double* lam_zimz = lambda_out + M;
      const int nk = skato_filter_eigen(ev, n, lam_zimz);
      if (nk < 0) {
        out->eig_ok[11] = 0;
      } else {
        out->zimz_nlambda = nk;
        double ls = 0, l2 = 0, l4 = 0;
        for (int i = 0; i < nk; ++i) {
          const double l = lam_zimz[i];
          ls += l;
          l2 += l * l;
          l4 += l * l * l * l;
        }
        out->zimz_lambda_sum = ls;
        out->muQ = ls;
        out->varQ = 2.0 * l2 + out->varZeta;
        const double KerQ = l4 / l2 / l2 * 12;
        out->df = 12 / KerQ;
        out->eig_ok[11] = 1;
      }
    } else {
      const int nk = skato_filter_eigen(ev, n, tmpv);
      if (nk < 0) {
        out->eig_ok[k] = 0;
      } else {
        const SkatoMoment mo = skato_moment(tmpv, nk);
        out->mom_mu[k] = mo.muQ;
        out->mom_var[k] = mo.varQ;
        out->mom_df[k] = mo.df;
        out->eig_ok[k] = 1;
      }
    }
  }
  co.sync();
}

// Stage B2, all 13 eigenproblems of a gene in ONE workgroup (round 5).  The per-problem form above gives every problem a
// workgroup of its own and every eigenvalue a lane: 13 waves per gene, each as slow as its slowest eigenvalue (~55 evaluations
// of the recurrence where the average eigenvalue needs ~25, rvt_coop.h).  Here the 13 scaled tridiagonals sit side by side in
// `vec` (3 m doubles per problem: d, e^2, eigenvalues) and the lanes walk the list of (problem, eigenvalue) tasks with a
// stride — a lane solves several eigenvalues of different problems one after the other, so what it pays is the AVERAGE cost.
// Every number is the per-problem form's: the same raw arrays, the same Gershgorin bounds and power-of-two scaling (computed
// by one thread per problem), the same sturm_eigenvalue, the same filters and moments.  `vec` >= 39 * m + 64 doubles.
struct SpectrumMeta {
  double lo, hi, span;
  int sh, n, active;
};
RVT_HD void gene_spectrum_all(const Coop& co, const NullConsts& nc, int M, int Mp, unsigned tests, GeneScratch ws,
                              double* vec, SpectrumMeta* meta, GeneStats* out, double* lambda_out) {
  const int m = out->n_poly;
  if (m == 0) return;
  if (m == 1) {  // the single-variant shortcuts: no recurrence to run
    gene_spectrum(co, nc, 11, M, Mp, tests, ws, vec, out, lambda_out);
    gene_spectrum(co, nc, 12, M, Mp, tests, ws, vec, out, lambda_out);
    return;
  }
  const double* d0 = ws.tri;
  const double* e0 = ws.tri + Mp;
  const bool shares = skat_shares_weights(ws, Mp, m);
  const bool zimz_bad = !(d0[0] > 0.0);  // 1'A1 <= 0: the reference's M is 0/0 and every eigenvalue NaN
  auto active = [&](int k) {
    if (k == 12) return (tests & RVT_TEST_SKAT) != 0;
    if (!(tests & RVT_TEST_SKATO)) return false;
    return !(k == 11 && zimz_bad);
  };
  auto size_of = [&](int k) { return k == 11 ? m - 1 : m; };
  // ---- A: the raw tridiagonals, one entry per item
  for (int item = co.tid; item < kNEigen * m; item += co.nt) {
    const int k = item / m, i = item % m;
    if (!active(k) || i >= size_of(k)) continue;
    double* td = vec + (size_t)k * 3 * m;
    double* te = td + m;
    if (k == 12) {
      if (shares) {
        const double sc = 2.0 * (nc.binary ? 1.0 : nc.sigma2);
        td[i] = sc * d0[i];
        te[i] = sc * e0[i];
      } else {
        td[i] = ws.tri[2 * Mp + i];
        te[i] = ws.tri[3 * Mp + i];
      }
    } else if (k == 11) {
      td[i] = (i == 0) ? d0[1] - e0[0] * e0[0] / d0[0] : d0[i + 1];
      te[i] = e0[i + 1];
    } else {
      const double rh = skato_rho_value(k);
      const double s1 = 1.0 - rh, sm = 1.0 - rh + (double)m * rh;
      const double s1m = sqrt(s1 * sm);
      td[i] = (i == 0) ? sm * d0[0] : s1 * d0[i];
      te[i] = (i == 0) ? s1m * e0[0] : s1 * e0[i];
    }
  }
  co.sync();
  // ---- B: per problem, one thread: Gershgorin interval, the power of two that brings its span into [1/2, 1)
  const double pivmin = DBL_MIN * 1024.0;
  for (int k = co.tid; k < kNEigen; k += co.nt) {
    SpectrumMeta mt;
    mt.active = active(k) ? 1 : 0;
    mt.n = size_of(k);
    mt.lo = mt.hi = mt.span = 0.0;
    mt.sh = 0;
    if (mt.active) {
      const double* d = vec + (size_t)k * 3 * m;
      const double* e = d + m;
      const int n = mt.n;
      double lo = d[0], hi = d[0];
      for (int j = 0; j < n; ++j) {
        const double r = (j > 0 ? fabs(e[j - 1]) : 0.0) + (j < n - 1 ? fabs(e[j]) : 0.0);
        lo = fmin(lo, d[j] - r);
        hi = fmax(hi, d[j] + r);
      }
      const double span0 = fmax(fabs(lo), fabs(hi));
      int sh = 0;
      if (span0 > 0.0 && span0 < INFINITY) (void)frexp(span0, &sh);
      lo = ldexp(lo, -sh);
      hi = ldexp(hi, -sh);
      const double span = fmax(fabs(lo), fabs(hi));
      mt.lo = lo - (2.0 * kDblEps * span * n + 2.0 * pivmin);
      mt.hi = hi + (2.0 * kDblEps * span * n + 2.0 * pivmin);
      mt.span = span;
      mt.sh = sh;
    }
    meta[k] = mt;
  }
  co.sync();
  // ---- C: scale (exact) and square the couplings, floored as coop_tridiag_eigvals floors them
  for (int item = co.tid; item < kNEigen * m; item += co.nt) {
    const int k = item / m, i = item % m;
    if (!meta[k].active || i >= meta[k].n) continue;
    double* td = vec + (size_t)k * 3 * m;
    double* te = td + m;
    td[i] = ldexp(td[i], -meta[k].sh);
    if (i < meta[k].n - 1) {
      const double es = ldexp(te[i], -meta[k].sh);
      te[i] = fmax(es * es, 0x1p-200);
    }
  }
  co.sync();
  // ---- C2: the rho problems, one thread each: moments from the traces of the tridiagonal's powers when the reference's
  //      filter provably drops rounding only (skato_moment_by_trace) — such a problem (active = 2) needs no eigenvalues
  for (int k = co.tid; k < 11; k += co.nt) {
    if (meta[k].active != 1) continue;
    const double* d = vec + (size_t)k * 3 * m;
    SkatoMoment mo;
    if (skato_moment_by_trace(d, d + m, meta[k].n, meta[k].sh, &mo)) {
      out->mom_mu[k] = mo.muQ;
      out->mom_var[k] = mo.varQ;
      out->mom_df[k] = mo.df;
      out->eig_ok[k] = 1;
      meta[k].active = 2;
    }
  }
  co.sync();
  // ---- D: the (problem, eigenvalue) tasks, strided over the lanes
  int off[kNEigen + 1];
  off[0] = 0;
  for (int k = 0; k < kNEigen; ++k) off[k + 1] = off[k] + (meta[k].active == 1 ? meta[k].n : 0);
  for (int t = co.tid; t < off[kNEigen]; t += co.nt) {
    int k = 0;
    while (t >= off[k + 1]) ++k;
    const int idx = t - off[k];
    const double* d = vec + (size_t)k * 3 * m;
    double* ev = vec + (size_t)k * 3 * m + 2 * m;
    ev[idx] = ldexp(sturm_eigenvalue(d, d + m, meta[k].n, idx, meta[k].lo, meta[k].hi, meta[k].span, pivmin), meta[k].sh);
  }
  co.sync();
  // ---- E: per problem, one thread: the reference's filters and moments (the kept values overwrite the problem's d)
  for (int k = co.tid; k < kNEigen; k += co.nt) {
    if (k == 12 ? !(tests & RVT_TEST_SKAT) : !(tests & RVT_TEST_SKATO)) continue;
    if (!meta[k].active) {
      out->eig_ok[k] = 0;  // (k == 11 with 1'A1 <= 0)
      continue;
    }
    if (meta[k].active == 2) continue;  // (stage C2 wrote the moments)
    const int n = meta[k].n;
    const double* ev = vec + (size_t)k * 3 * m + 2 * m;
    double* tmpv = vec + (size_t)k * 3 * m;
    if (k == 12) {
      double* lam_skat = lambda_out;
      const int r_ub = (nc.N < (int64_t)m) ? (int)nc.N : m;
      int r = 0;
      for (int i = n - 1; i >= 0; --i) {
        if (ev[i] > 1e-30 && r < r_ub) {
          lam_skat[r++] = ev[i];
        } else
          break;
      }
      out->skat_nlambda = r;
      out->eig_ok[12] = 1;
    } else if (k == 11) {
      double* lam_zimz = lambda_out + M;
      const int nk = skato_filter_eigen(ev, n, lam_zimz);
      if (nk < 0) {
        out->eig_ok[11] = 0;
      } else {
        out->zimz_nlambda = nk;
        double ls = 0, l2 = 0, l4 = 0;
        for (int i = 0; i < nk; ++i) {
          const double l = lam_zimz[i];
          ls += l;
          l2 += l * l;
          l4 += l * l * l * l;
        }
        out->zimz_lambda_sum = ls;
        out->muQ = ls;
        out->varQ = 2.0 * l2 + out->varZeta;
        const double KerQ = l4 / l2 / l2 * 12;
        out->df = 12 / KerQ;
        out->eig_ok[11] = 1;
      }
    } else {
      const int nk = skato_filter_eigen(ev, n, tmpv);
      if (nk < 0) {
        out->eig_ok[k] = 0;
      } else {
        const SkatoMoment mo = skato_moment(tmpv, nk);
        out->mom_mu[k] = mo.muQ;
        out->mom_var[k] = mo.varQ;
        out->mom_df[k] = mo.df;
        out->eig_ok[k] = 1;
      }
    }
  }
  co.sync();
}

// did SkatO::Fit succeed?  (every getEigen found a positive eigenvalue)
RVT_HD bool skato_fit_ok(const GeneStats& gs) {
  if (!gs.skato_ok || gs.n_poly == 0) return false;
  if (gs.skato_single) return gs.eig_ok[11] != 0;
  for (int k = 0; k < 12; ++k)
    if (!gs.eig_ok[k]) return false;
  return true;
}

}  // namespace rvt
