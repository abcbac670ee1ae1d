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

struct GeneScratch {
  double* R;     // Mp x Cp   reduced statistics, row-major (ld = Cp)
  double* A;     // Mmax x Mmax
  double* B;     // Mmax x Mmax   (work copy for the eigen solver)
  double* suf;   // Mmax x Mmax   row suffix sums of A
  double* vecs;  // 16 * Mmax doubles of vector scratch
  int* ivec;     // 2 * Mmax ints
};
RVT_HD size_t gene_scratch_doubles(int Mp, int Cp) {
  return (size_t)Mp * Cp + 3 * (size_t)Mp * Mp + 16 * (size_t)Mp + (size_t)Mp;  // ivec packed in the tail
}
RVT_HD GeneScratch gene_scratch_carve(double* mem, int Mp, int Cp) {
  GeneScratch s;
  s.R = mem;
  s.A = s.R + (size_t)Mp * Cp;
  s.B = s.A + (size_t)Mp * Mp;
  s.suf = s.B + (size_t)Mp * Mp;
  s.vecs = s.suf + (size_t)Mp * Mp;
  s.ivec = (int*)(s.vecs + 16 * (size_t)Mp);
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

// The whole per-gene statistics stage.
//   parts:   P partial matrices of Mp x Cp doubles (row-major); only tiles with tile_col >= tile_row hold data
//   colstat: P x 3 x Mp  (sum, min, max) partials
//   bparts:  PB x 2 x burden_rec_len(d) partial burden sums (CMC, Zeggini) — may be null when no burden test
//   lambda_out: 2*M doubles: [0,M) SKAT eigenvalues, [M,2M) Z(I-M)Z' eigenvalues
RVT_HD void gene_stats(const Coop& co, const NullConsts& nc, int M, int Mp, int Cp, const double* parts, int P,
                       const double* colstat, const double* bparts, int PB, const double* af, const rvt_params& prm,
                       unsigned tests, GeneScratch ws, GeneStats* out, double* lambda_out, int* flip_out,
                       int* kept_out) {
  const int d = nc.d;
  const int ldr = Cp;
  double* R = ws.R;
  // ---- 1. reduce the partial statistics (fixed order => deterministic) ----------------------------
  for (int idx = co.tid; idx < Mp * Cp; idx += co.nt) {
    const int i = idx / Cp, j = idx % Cp;
    double s = 0.0;
    if ((j >> 4) >= (i >> 4)) {
      for (int p = 0; p < P; ++p) s += parts[(size_t)p * Mp * Cp + idx];
    }
    R[idx] = s;
  }
  double* colsum = ws.vecs;            // [Mp]
  double* cmin = ws.vecs + Mp;         // [Mp]
  double* cmax = ws.vecs + 2 * Mp;     // [Mp]
  double* sgn = ws.vecs + 3 * Mp;      // [Mp]  +1 / -1
  double* shf = ws.vecs + 4 * Mp;      // [Mp]   0 / 2
  double* bw = ws.vecs + 5 * Mp;       // beta weights (filtered index)
  double* ut = ws.vecs + 6 * Mp;       // weighted scores
  double* rowsum = ws.vecs + 7 * Mp;   // A·1
  double* ev = ws.vecs + 8 * Mp;       // eigenvalues ascending
  double* td = ws.vecs + 9 * Mp;       // tridiagonal d
  double* te = ws.vecs + 10 * Mp;      // tridiagonal e
  double* hv = ws.vecs + 11 * Mp;      // householder v
  double* hw = ws.vecs + 12 * Mp;      // householder w
  double* cd = ws.vecs + 13 * Mp;      // cholesky diag of R_rho
  double* cc = ws.vecs + 14 * Mp;      // cholesky below-diagonal constant per column
  double* tmpv = ws.vecs + 15 * Mp;
  int* kidx = ws.ivec;                 // kept column list
  for (int j = co.tid; j < M; j += co.nt) {
    double s = 0.0, mn = INFINITY, mx = -INFINITY;
    for (int p = 0; p < P; ++p) {
      const double* c = colstat + (size_t)p * 3 * Mp;
      s += c[j];
      mn = fmin(mn, c[Mp + j]);
      mx = fmax(mx, c[2 * Mp + j]);
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
  // symmetric completion of the G'DG block
  for (int idx = co.tid; idx < M * M; idx += co.nt) {
    const int i = idx / M, j = idx % M;
    if (j < i) R[(size_t)i * ldr + j] = R[(size_t)j * ldr + i];  // lower triangle := upper (exact symmetry)
  }
  co.sync();
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
    out->status = (m == 0) ? RVT_ST_NO_POLY : 0;
    out->skato_ok = 0;
    out->skato_single = 0;
    out->skat_nlambda = 0;
    out->zimz_nlambda = 0;
    out->skat_lambda_off = 0;
    out->zimz_lambda_off = M;
    out->cmc_ok = out->zeg_ok = 0;
    out->cmc_nonref = 0;
  }
  co.sync();
  const int m = kidx[Mp];
  // ---- 8. burden score statistics (independent of the rest) ----------------------------------------
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
  if (m == 0 || !(tests & (RVT_TEST_SKAT | RVT_TEST_SKATO))) {
    co.sync();
    return;
  }
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
  // ---- 4. projected matrix  Wm = S' − T' Cinv T'ᵀ  on the kept columns (into A, column-major m x m) --
  double* A = ws.A;
  for (int idx = co.tid; idx < m * m; idx += co.nt) {
    const int a = idx % m, b = idx / m;
    const int ia = kidx[a], ib = kidx[b];
    double q = 0.0;
    for (int k = 0; k < d; ++k) {
      double s = 0.0;
      for (int l = 0; l < d; ++l) s += nc.Cinv[k * d + l] * R[(size_t)ib * ldr + M + l];
      q += R[(size_t)ia * ldr + M + k] * s;
    }
    A[(size_t)b * m + a] = R[(size_t)ia * ldr + ib] - q;
  }
  co.sync();
  const double vscale = nc.binary ? 1.0 : nc.sigma2;  // quantitative: statistics were unweighted, v = sigma2
  double* lam_skat = lambda_out;
  double* lam_zimz = lambda_out + M;
  // ---- 5/6. SKAT --------------------------------------------------------------------------------------
  if (tests & RVT_TEST_SKAT) {
    for (int a = co.tid; a < m; a += co.nt) {
      double freq = af[a];  // quirk: filtered position a reads the counter of unfiltered column a
      if (freq > 0.5) freq = 1.0 - freq;
      double wgt = 0.0;
      if (freq > 1e-30) {
        wgt = beta_density(freq, prm.skat_beta1, prm.skat_beta2);
        wgt *= wgt;
      }
      bw[a] = sqrt(wgt);
      const double s = bw[a] * R[(size_t)kidx[a] * ldr + M + d];
      ut[a] = s * s;
    }
    co.sync();
    double* Bm = ws.B;
    for (int idx = co.tid; idx < m * m; idx += co.nt) {
      const int a = idx % m, b = idx / m;
      Bm[idx] = bw[a] * (vscale * A[idx]) * bw[b];
    }
    co.sync();
    coop_sym_eigvals(co, Bm, m, td, te, hv, hw, ev);
    if (co.tid == 0) {
      double Q = 0.0;
      for (int a = 0; a < m; ++a) Q += ut[a];
      out->skat_Q = Q;
      const int r_ub = (nc.N < (int64_t)m) ? (int)nc.N : m;
      int r = 0;
      for (int i = m - 1; i >= 0; --i) {
        if (ev[i] > 1e-30 && r < r_ub) {
          lam_skat[r++] = ev[i];
        } else
          break;
      }
      out->skat_nlambda = r;
    }
    co.sync();
  }
  if (!(tests & RVT_TEST_SKATO)) return;
  // ---- 7. SKAT-O ----------------------------------------------------------------------------------------
  for (int a = co.tid; a < m; a += co.nt) {
    double freq = af[a];
    if (freq > 0.5) freq = 1.0 - freq;
    bw[a] = (freq > 1e-30) ? beta_density(freq, prm.skato_beta1, prm.skato_beta2) : 0.0;
    ut[a] = bw[a] * R[(size_t)kidx[a] * ldr + M + d];
  }
  co.sync();
  // A <- B Wm B / 2   (= Z1'Z1)
  for (int idx = co.tid; idx < m * m; idx += co.nt) {
    const int a = idx % m, b = idx / m;
    A[idx] = bw[a] * A[idx] * bw[b] / 2.0;
  }
  co.sync();
  double s2;
  if (nc.binary)
    s2 = 1.0;
  else {
    s2 = sqrt(nc.rss);
    s2 = (s2 * s2) / (double)(nc.N - 1);
  }
  double su = 0.0, su2 = 0.0;
  for (int a = 0; a < m; ++a) {  // every thread: m is small
    su += ut[a];
    su2 += ut[a] * ut[a];
  }
  if (m == 1) {
    // FitSKAT: Q = u²/s2/2, W = A, Davies on its single eigenvalue (=> Liu)
    if (co.tid == 0) {
      out->skato_single = 1;
      double Q = ut[0] * ut[0];
      if (!nc.binary) Q /= nc.rss / (double)(nc.N - 1);  // FitSKAT: squaredNorm()/(nPeople-1)
      Q /= 2.;
      out->Qs[0] = Q;
      const double lam = A[0];
      if (lam > 0) {
        lam_zimz[0] = lam;
        out->zimz_nlambda = 1;
        out->skato_ok = 1;
      } else {
        out->skato_ok = 0;
        out->status |= RVT_ST_SKATO_EIGEN;
      }
    }
    co.sync();
    return;
  }
  double rho[kNRho];
  for (int i = 0; i < kNRho; ++i) {
    const double r0 = 1.0 * i / 10;
    rho[i] = (r0 > 0.999) ? 0.999 : r0;
  }
  // row suffix sums of A: suf[i][q] = sum_{j >= q} A[i][j]      (rho independent)
  double* suf = ws.suf;
  for (int i = co.tid; i < m; i += co.nt) {
    double s = 0.0;
    for (int q = m - 1; q >= 0; --q) {
      s += A[(size_t)q * m + i];
      suf[(size_t)q * m + i] = s;
    }
    rowsum[i] = s;
  }
  co.sync();
  int ok = 1;
  for (int ir = 0; ir < kNRho && ok; ++ir) {
    const double rh = rho[ir];
    // Cholesky factor of R_rho = (1-rho) I + rho 11': L[j][j] = cd[j], L[i][j] = cc[j] (i > j)
    if (co.tid == 0) {
      double acc = 0.0;  // sum_{k<j} cc[k]^2
      for (int j = 0; j < m; ++j) {
        const double dj = sqrt(1.0 - acc);
        cd[j] = dj;
        cc[j] = (rh - acc) / dj;
        acc += cc[j] * cc[j];
      }
    }
    co.sync();
    // AL[i][q] = A[i][q] cd[q] + cc[q] * suf[i][q+1]   -> into B (column-major)
    double* Bm = ws.B;
    for (int idx = co.tid; idx < m * m; idx += co.nt) {
      const int i = idx % m, q = idx / m;
      const double tail = (q + 1 < m) ? suf[(size_t)(q + 1) * m + i] : 0.0;
      Bm[idx] = A[idx] * cd[q] + cc[q] * tail;
    }
    co.sync();
    // K[p][q] = cd[p] AL[p][q] + cc[p] * sum_{i > p} AL[i][q]   (column suffix sums), in place per column
    for (int q = co.tid; q < m; q += co.nt) {
      double tail = 0.0;  // sum_{i > p} AL[i][q]
      for (int p = m - 1; p >= 0; --p) {
        const double alpq = Bm[(size_t)q * m + p];
        Bm[(size_t)q * m + p] = cd[p] * alpq + cc[p] * tail;
        tail += alpq;
      }
    }
    co.sync();
    // symmetrise (rounding) so the eigen solver sees an exactly symmetric matrix
    for (int idx = co.tid; idx < m * m; idx += co.nt) {
      const int i = idx % m, j = idx / m;
      if (i > j) {
        const double s = 0.5 * (Bm[(size_t)j * m + i] + Bm[(size_t)i * m + j]);
        Bm[(size_t)j * m + i] = s;
      }
    }
    co.sync();
    for (int idx = co.tid; idx < m * m; idx += co.nt) {
      const int i = idx % m, j = idx / m;
      if (i < j) Bm[(size_t)j * m + i] = Bm[(size_t)i * m + j];
    }
    co.sync();
    coop_sym_eigvals(co, Bm, m, td, te, hv, hw, ev);
    if (co.tid == 0) {
      const int nk = skato_filter_eigen(ev, m, tmpv);
      if (nk < 0) {
        kidx[Mp + 1] = 0;
      } else {
        kidx[Mp + 1] = 1;
        const SkatoMoment mo = skato_moment(tmpv, nk);
        out->mom_mu[ir] = mo.muQ;
        out->mom_var[ir] = mo.varQ;
        out->mom_df[ir] = mo.df;
        double q = (1.0 - rh) * su2 + rh * (su * su);
        q /= s2;
        q /= 2.0;
        out->Qs[ir] = q;
      }
    }
    co.sync();
    ok = kidx[Mp + 1];
    co.sync();
  }
  if (!ok) {
    if (co.tid == 0) {
      out->skato_ok = 0;
      out->status |= RVT_ST_SKATO_EIGEN;
    }
    co.sync();
    return;
  }
  // Z(I-M)Z' = A − (A1)(A1)'/(1'A1)
  double tot = 0.0, r2 = 0.0;
  for (int a = 0; a < m; ++a) {
    tot += rowsum[a];
    r2 += rowsum[a] * rowsum[a];
  }
  double* Bm = ws.B;
  double vzpart = 0.0;
  for (int idx = co.tid; idx < m * m; idx += co.nt) {
    const int i = idx % m, j = idx / m;
    const double zmz = rowsum[i] * rowsum[j] / tot;
    const double zimz = A[idx] - zmz;
    Bm[idx] = zimz;
    vzpart += zmz * zimz;
  }
  const double vz = co.sum(vzpart);
  co.sync();
  coop_sym_eigvals(co, Bm, m, td, te, hv, hw, ev);
  if (co.tid == 0) {
    const int nk = skato_filter_eigen(ev, m, lam_zimz);
    if (nk < 0) {
      out->skato_ok = 0;
      out->status |= RVT_ST_SKATO_EIGEN;
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
      out->varZeta = 4.0 * vz;
      out->muQ = ls;
      out->varQ = 2.0 * l2 + out->varZeta;
      const double KerQ = l4 / l2 / l2 * 12;
      out->df = 12 / KerQ;
      // tau_rho = m² rho z_norm + (1-rho) ||z̄'Z1||² / z_norm, z_norm = 1'A1/m², z̄'Z1 = (A1)'/m
      const double z_norm = tot / ((double)m * (double)m);
      const double zz = r2 / ((double)m * (double)m);
      for (int i = 0; i < kNRho; ++i)
        out->tau[i] = (double)(m * m) * rho[i] * z_norm + (1.0 - rho[i]) * zz / z_norm;
      out->skato_ok = 1;
    }
  }
  co.sync();
}

}  // namespace rvt
