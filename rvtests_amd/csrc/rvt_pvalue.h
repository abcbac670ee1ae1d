// rvtests_amd — per-gene p-value stage, serial form.
//
// `gene_pvalue_serial` walks the p-value stage for one gene on ONE thread.  The GPU kernel
// (gene_pvalue_kernel in pvalue_kernels.hip) runs the same steps with the independent pieces — the 11
// per-rho tail probabilities, the 11 quantiles, and the 21/42 quadrature abscissae of every QAGS step —
// spread over the lanes of one wave; this serial form is what the host test harness calls, and what
// documents the order of operations:
//   SKAT    Skat.cpp:100-103      Davies, Liu when p <= 0 or p == 1
//   SKAT-O  SkatO.cpp:206-277     p_rho by moments, min-p, Q_minP, QAGS of the Davies integrand on
//                                 [0,40] (epsabs 1e-25, epsrel 1.220703e-4, limit 1000), Liu integrand
//                                 when that fails, corrections; FitSKAT for a single variant (:60-99)
//   burden  LinearRegressionScoreTest.cpp:259-261   1-df chi-square tail
#pragma once
#include "rvt_gene.h"

namespace rvt {

constexpr double kSkatoEpsAbs = 1e-25;
constexpr double kSkatoEpsRel = 0.0001220703;
constexpr int kSkatoLimit = 1000;

RVT_HD void pvalue_init_result(const GeneStats& gs, int64_t gene_id, rvt_gene_result* r) {
  r->gene_id = gene_id;
  r->status = (uint32_t)gs.status;
  if (gs.n_poly > 0 && gs.skato_ok && !skato_fit_ok(gs)) r->status |= RVT_ST_SKATO_EIGEN;
  r->n_variants = gs.n_variants;
  r->n_poly = gs.n_poly;
  r->skat_ok = 0;
  r->skat_Q = r->skat_p = 0.0;
  r->skat_nlambda = gs.skat_nlambda;
  r->perm_ok = 0;
  r->perm_num_perm = r->perm_actual_perm = r->perm_num_greater = r->perm_num_equal = 0;
  r->perm_pvalue = 0.0;
  r->famskat_ok = 0;
  r->famcmc_ok = r->famzeg_ok = 0;
  r->famcmc_af = r->famcmc_U = r->famcmc_V = r->famcmc_p = 0.0;
  r->famzeg_af = r->famzeg_U = r->famzeg_V = r->famzeg_p = 0.0;
  r->famskat_Q = r->famskat_p = 0.0;
  r->skato_ok = 0;
  r->skato_Q = r->skato_rho = r->skato_p = 0.0;
  r->skato_qags_status = 0;
  r->skato_qags_neval = 0;
  r->cmc_ok = gs.cmc_ok;
  r->cmc_nonref = gs.cmc_nonref;
  r->cmc_U = gs.cmc_U;
  r->cmc_V = gs.cmc_V;
  r->cmc_stat = gs.cmc_stat;
  r->cmc_p = 0.0;
  r->zeg_ok = gs.zeg_ok;
  r->zeg_U = gs.zeg_U;
  r->zeg_V = gs.zeg_V;
  r->zeg_stat = gs.zeg_stat;
  r->zeg_p = 0.0;
  r->davies_terms = 0.0;
}

// SkatO bookkeeping between the per-rho p-values and the integral: min-p, rho, Q (SkatO.cpp:216-233)
RVT_HD void skato_select(const GeneStats& gs, const double* pvals, double* minP_out, int* minIndex_out) {
  double minP = pvals[0];
  int minIndex = 0;
  for (int i = 1; i < kNRho; ++i)
    if (pvals[i] < minP) {
      minP = pvals[i];
      minIndex = i;
    }
  *minP_out = minP;
  *minIndex_out = minIndex;
}

RVT_HD void skato_fill_integrand(const GeneStats& gs, const double* qminp, const double* lam, const int* th,
                                 SkatoIntegrand* s) {
  for (int i = 0; i < kNRho; ++i) {
    const double r0 = 1.0 * i / 10;
    s->rho[i] = (r0 > 0.999) ? 0.999 : r0;
    s->qminp[i] = qminp[i];
    s->tau[i] = gs.tau[i];
  }
  s->muQ = gs.muQ;
  s->varQ = gs.varQ;
  s->varZeta = gs.varZeta;
  s->df = gs.df;
  s->lambda = lam;
  s->th = th;
  s->r = gs.zimz_nlambda;
  s->lambda_sum = gs.zimz_lambda_sum;
  s->pre = nullptr;
  s->liu = nullptr;
  s->lg_half = lgamma(0.5);
}

// th_skat / th_zimz: scratch of >= n ints each; qags_mem: qags_workspace_bytes(kSkatoLimit) bytes
RVT_HD void gene_pvalue_serial(const GeneStats& gs, const double* lambda_buf, unsigned tests, int64_t gene_id,
                               int* th_skat, int* th_zimz, void* qags_mem, rvt_gene_result* r) {
  pvalue_init_result(gs, gene_id, r);
  if (gs.n_poly == 0) return;
  double terms = 0.0, nt;
  if (tests & RVT_TEST_SKAT) {
    const double* lam = lambda_buf + gs.skat_lambda_off;
    const int n = gs.skat_nlambda;
    davies_order(lam, n, th_skat);
    int fault;
    double p = davies_pvalue(lam, th_skat, n, gs.skat_Q, &fault, &nt);
    terms += nt;
    if (p <= 0.0 || p == 1.0) p = liu_pvalue(lam, n, gs.skat_Q);
    r->skat_ok = 1;
    r->skat_Q = gs.skat_Q;
    r->skat_p = p;
  }
  if ((tests & RVT_TEST_SKATO) && skato_fit_ok(gs)) {
    const double* lam = lambda_buf + gs.zimz_lambda_off;
    const int n = gs.zimz_nlambda;
    davies_order(lam, n, th_zimz);
    if (gs.skato_single) {
      int fault;
      r->skato_Q = gs.Qs[0];
      r->skato_rho = 0.0;
      r->skato_p = davies_pvalue(lam, th_zimz, n, gs.Qs[0], &fault, &nt);
      r->skato_ok = 1;
    } else {
      double pvals[kNRho], qminp[kNRho], minP;
      int minIndex;
      SkatoMoment mo[kNRho];
      for (int i = 0; i < kNRho; ++i) {
        mo[i].muQ = gs.mom_mu[i];
        mo[i].varQ = gs.mom_var[i];
        mo[i].df = gs.mom_df[i];
        pvals[i] = skato_p_by_moment(gs.Qs[i], mo[i]);
      }
      skato_select(gs, pvals, &minP, &minIndex);
      for (int i = 0; i < kNRho; ++i) qminp[i] = skato_q_by_moment(minP, mo[i]);
      SkatoIntegrand si;
      skato_fill_integrand(gs, qminp, lam, th_zimz, &si);
      DaviesPrelude pre;
      DaviesMemo memo;  // the coefficient sums of the searches, shared by every abscissa of the quadrature (rvt_davies.h)
      dv_memo_clear((dv_memo_p)&memo);
      davies_prelude(lam, th_zimz, n, 10000, 0.000001, &pre, true, &memo);
      si.pre = &pre;
      const LiuPre liu = liu_prepare(lam, n);
      si.liu = &liu;
      QagsWorkspace ws = qags_workspace_carve(qags_mem, kSkatoLimit);
      double fv[42];
      int neval = 0, status;
      double integral;
      for (int pass = 0; pass < 2; ++pass) {
        QagsMachine qm;
        qm.begin(0., 40., kSkatoEpsAbs, kSkatoEpsRel, kSkatoLimit, ws);
        if (qm.running()) {
          for (int t = 0; t < 21; ++t) {
            const double x = gk21_abscissa(0., 40., t);
            fv[t] = pass == 0 ? skato_integrand_davies(si, x, &nt) : skato_integrand_liu(si, x);
            if (pass == 0) terms += nt;
          }
          neval += 21;
          qm.first_panel(fv);
        }
        while (qm.running()) {
          double a1, b1, b2;
          qm.bisect(&a1, &b1, &b2);
          for (int t = 0; t < 42; ++t) {
            const double x = (t < 21) ? gk21_abscissa(a1, b1, t) : gk21_abscissa(b1, b2, t - 21);
            fv[t] = pass == 0 ? skato_integrand_davies(si, x, &nt) : skato_integrand_liu(si, x);
            if (pass == 0) terms += nt;
          }
          neval += 42;
          qm.advance(fv, fv + 21);
        }
        integral = qm.result;
        status = qm.status;
        if (pass == 0) {
          r->skato_qags_status = status;
          if (status == 0) break;
        } else {
          r->skato_qags_status = r->skato_qags_status * 100 + status;
        }
      }
      r->skato_qags_neval = neval;
      double rho = (minIndex == 10) ? 0.999 : 1.0 * minIndex / 10;
      if (rho >= 0.999) rho = 1.;
      r->skato_rho = rho;
      r->skato_Q = gs.Qs[minIndex];
      r->skato_p = skato_finish(integral, minP, pvals);
      r->skato_ok = 1;
    }
  }
  if ((tests & RVT_TEST_CMC) && gs.cmc_ok) r->cmc_p = chisq_Q(gs.cmc_stat, 1.0);
  if ((tests & RVT_TEST_ZEGGINI) && gs.zeg_ok) r->zeg_p = chisq_Q(gs.zeg_stat, 1.0);
  r->davies_terms = terms;
}

}  // namespace rvt
