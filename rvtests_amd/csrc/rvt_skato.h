// rvtests_amd — SKAT-O optimal-rho p-value pieces shared by the p-value kernel and its host harness.
//
// Follows regression/SkatO.cpp: per-rho moment matching :383-435, min-p and Q_minP :216-233, the
// Davies integrand with Liu fallback :303-325, the Liu integrand :327-337, and the post-integration
// corrections :258-277.
#pragma once
#include "rvt_davies.h"
#include "rvt_qags.h"

namespace rvt {

constexpr int kNRho = 11;

struct SkatoMoment {
  double muQ, varQ, df;
};

// SkatOImpl::getMoment       (regression/SkatO.cpp:383-418)
// (from the power sums c0..c3 = sum lambda^1..4)
RVT_HD SkatoMoment skato_moment_from_sums(double c0, double c1, double c2, double c3) {
  SkatoMoment m;
  m.muQ = c0;
  const double sigmaQ = sqrt(2 * c1);
  const double s1 = c2 / c1 / sqrt(c1);
  const double s2 = c3 / (c1 * c1);
  double l;
  if (s1 * s1 > s2) {
    const double a = 1 / (s1 - sqrt(s1 * s1 - s2));
    const double d = (s1 * a - 1.0 * a * a);
    l = a * a - 2 * d;
  } else {
    l = 1. / s2;
  }
  m.varQ = sigmaQ * sigmaQ;
  m.df = l;
  return m;
}
RVT_HD SkatoMoment skato_moment(const double* la, int n) {
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (int i = 0; i < n; ++i) {
    const double l = la[i], l2 = l * l;
    c0 += l;
    c1 += l2;
    c2 += l2 * l;
    c3 += l2 * l2;
  }
  return skato_moment_from_sums(c0, c1, c2, c3);
}

// getPvalByMoment / getQvalByMoment       (regression/SkatO.cpp:420-435)
RVT_HD double skato_p_by_moment(double Q, const SkatoMoment& m) {
  const double Q_Norm = (Q - m.muQ) / sqrt(m.varQ) * sqrt(2. * m.df) + m.df;
  return chisq_Q(Q_Norm, m.df);
}
RVT_HD double skato_q_by_moment(double min_pval, const SkatoMoment& m) {
  const double q_org = chisq_quantile_Q(min_pval, m.df);
  return (q_org - m.df) / sqrt(2. * m.df) * sqrt(m.varQ) + m.muQ;
}

// Everything the integrand needs (per gene)
struct SkatoIntegrand {
  double rho[kNRho];    // capped at 0.999
  double qminp[kNRho];  // Q_minP per rho
  double tau[kNRho];
  double muQ, varQ, varZeta, df;
  const double* lambda;  // eigenvalues of Z(I-M)Z', descending, filtered
  const int* th;         // davies_order(lambda)
  int r;
  double lambda_sum;
  const DaviesPrelude* pre;  // c-independent part of qf() for `lambda` (may be null)
  const LiuPre* liu;         // coefficient-only part of Liu's approximation (may be null)
  double lg_half;            // lgamma(0.5) for the chi-square(1) density
};

RVT_HD double skato_kappa(const SkatoIntegrand& s, double x) {
  double kappa = DBL_MAX;
  for (int i = 0; i < kNRho; ++i) {
    const double v = (s.qminp[i] - s.tau[i] * x) / (1.0 - s.rho[i]);
    if (i == 0) kappa = v;
    if (v < kappa) kappa = v;
  }
  return kappa;
}

// integrandDavies      (regression/SkatO.cpp:303-325)
RVT_HD double skato_integrand_davies(const SkatoIntegrand& s, double x, double* nterms) {
  if (nterms) *nterms = 0.0;
  const double kappa = skato_kappa(s, x);
  double temp;
  if (kappa > s.lambda_sum * 10000) {
    temp = 0.0;
  } else {
    const double Q = (kappa - s.muQ) * sqrt(s.varQ - s.varZeta) / sqrt(s.varQ) + s.muQ;
    int fault;
    temp = davies_pvalue(s.lambda, s.th, s.r, Q, &fault, nterms, s.pre);
    if (temp <= 0.0 || temp == 1.0) temp = s.liu ? liu_pvalue_pre(*s.liu, Q) : liu_pvalue(s.lambda, s.r, Q);
  }
#if defined(RVT_DV_PROFILE) && !defined(__HIP_DEVICE_COMPILE__)
  rvt_dv_eval_mark();  // (profiling build of the host harness: one log record per abscissa)
#endif
  return (1.0 - temp) * chisq_density_lg(x, 1.0, s.lg_half);
}

// integrandLiu      (regression/SkatO.cpp:327-337)
RVT_HD double skato_integrand_liu(const SkatoIntegrand& s, double x) {
  double kappa = DBL_MAX;
  for (int i = 0; i < kNRho; ++i) {
    const double v = (s.qminp[i] - s.tau[i] * x) / (1.0 - s.rho[i]);
    if (v < kappa) kappa = v;
  }
  const double Q = (kappa - s.muQ) / sqrt(s.varQ) * sqrt(2.0 * s.df) + s.df;
  return chisq_P(Q, s.df) * chisq_density_lg(x, 1.0, s.lg_half);
}

// corrections after the integral      (regression/SkatO.cpp:258-277), nRho = 11 -> multi = 3
RVT_HD double skato_finish(double integral, double minP, const double* pvals) {
  double pValue = 1.0 - integral;
  if (pValue <= 0) {
    const double p = minP * 3;
    if (pValue < p) pValue = p;
  }
  if (pValue == 0.0) {
    pValue = pvals[0];
    for (int i = 1; i < kNRho; ++i)
      if (pvals[i] > 0 && pvals[i] < pValue) pValue = pvals[i];
  }
  return pValue;
}

}  // namespace rvt
