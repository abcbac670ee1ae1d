// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// CPU restatement of the reference's mixture-of-chi-square p-values:
//   MixtureChiSquare::getPvalue     regression/MixtureChiSquare.cpp:7-29   (Davies, fallback rules)
//   MixtureChiSquare::getLiuPvalue  regression/MixtureChiSquare.cpp:44-83  (Liu moment matching)
//   cdfchn(which=1)                 regression/cdflib.cpp:2634-2800        (argument checks)
//   cumchn                          regression/cdflib.cpp:5172-5350        (Poisson-weighted series,
//                                   eps = 1e-5, ntired = 1000 — kept verbatim: it fixes the digits)
//   cumchi -> cumgam -> gamma_inc   regression/cdflib.cpp:5141-5170,5581-5624,7892 (TOMS 654 GRATIO);
//                                   restated through the regularised incomplete gamma of
//                                   orc_special.cpp (agrees with GRATIO to ~1e-14, checked against
//                                   the compiled reference fragment in tests/test_oracle_ref.py).
#include <cmath>
#include "orc_api.h"

namespace {

void cumchi(double x, double df, double* cum, double* ccum) {
  const double a = df * 0.5, xx = x * 0.5;
  if (xx <= 0.0) {  // cumgam: cdflib.cpp:5611-5615
    *cum = 0.0;
    *ccum = 1.0;
    return;
  }
  *cum = orc_gamma_inc_P(a, xx);
  *ccum = orc_gamma_inc_Q(a, xx);
}

void cumchn(double x, double df, double pnonc, double* cum, double* ccum) {
  const double eps = 1.0e-5;
  const int ntired = 1000;
  auto dg = [&](int i) { return df + 2.0 * (double)i; };
  if (x <= 0.0) {
    *cum = 0.0;
    *ccum = 1.0;
    return;
  }
  if (pnonc <= 1.0e-10) {
    cumchi(x, df, cum, ccum);
    return;
  }
  const double xnonc = pnonc / 2.0;
  int icent = (int)(long)xnonc;  // fifidint: truncation
  if (icent == 0) icent = 1;
  const double chid2 = x / 2.0;
  double lfact = std::lgamma((double)(icent + 1));
  const double lcntwt = -xnonc + (double)icent * std::log(xnonc) - lfact;
  const double centwt = std::exp(lcntwt);
  double pcent, dummy;
  cumchi(x, dg(icent), &pcent, &dummy);
  double dfd2 = dg(icent) / 2.0;
  lfact = std::lgamma(1.0 + dfd2);
  const double lcntaj = dfd2 * std::log(chid2) - chid2 - lfact;
  const double centaj = std::exp(lcntaj);
  double sum = centwt * pcent;
  auto qsmall = [&](double xx) { return sum < 1.0e-20 || xx < eps * sum; };
  // backward
  int iterb = 0;
  double sumadj = 0.0, adj = centaj, wt = centwt, term = 0.0, pterm;
  int i = icent;
  for (;;) {
    dfd2 = dg(i) / 2.0;
    adj = adj * dfd2 / chid2;
    sumadj = sumadj + adj;
    pterm = pcent + sumadj;
    wt *= ((double)i / xnonc);
    term = wt * pterm;
    sum = sum + term;
    i -= 1;
    iterb = iterb + 1;
    if (iterb > ntired || qsmall(term) || i == 0) break;
  }
  // forward
  int iterf = 0;
  sumadj = adj = centaj;
  wt = centwt;
  i = icent;
  for (;;) {
    wt *= (xnonc / (double)(i + 1));
    pterm = pcent - sumadj;
    term = wt * pterm;
    sum = sum + term;
    i = i + 1;
    dfd2 = dg(i) / 2.0;
    adj = adj * chid2 / dfd2;
    sumadj = sum + adj;  // sic: the reference writes sumadj = sum + adj (cdflib.cpp:5340)
    iterf = iterf + 1;
    if (iterf > ntired || qsmall(term)) break;
  }
  *cum = sum;
  *ccum = 0.5 + (0.5 - *cum);
}

double sum_pow(const double* d, int n, int power) {
  double r = 0.0;
  for (int i = 0; i < n; ++i) {
    double tmp = d[i];
    for (int j = 1; j < power; ++j) tmp *= d[i];
    r += tmp;
  }
  return r;
}

}  // namespace

extern "C" {

// cumchn exposed for the fixture checks
void orc_cumchn(double x, double df, double pnonc, double* cum, double* ccum) { cumchn(x, df, pnonc, cum, ccum); }

// MixtureChiSquare::getLiuPvalue  (MixtureChiSquare.cpp:44-83)
double orc_liu_pvalue(const double* lambda, int n, double Q) {
  const double c1 = sum_pow(lambda, n, 1), c2 = sum_pow(lambda, n, 2), c3 = sum_pow(lambda, n, 3),
               c4 = sum_pow(lambda, n, 4);
  const double s1 = c3 / c2 / std::sqrt(c2), s2 = c4 / c2 / c2;
  const double muQ = c1, sigmaQ = std::sqrt(2.0 * c2), tstar = (Q - muQ) / sigmaQ;
  double a, delta, l;
  if (s1 * s1 > s2) {
    a = 1 / (s1 - std::sqrt(s1 * s1 - s2));
    delta = (s1 * a - 1) * a * a;
    l = a * a - 2.0 * delta;
  } else {
    a = 1.0 / s1;
    delta = 0.0;
    l = c2 * c2 * c2 / c3 / c3;
  }
  const double muX = l + delta, sigmaX = std::sqrt(2) * a;
  const double x = tstar * sigmaX + muX;
  // cdfchn(which = 1) argument checks: x < 0 -> status -4, df <= 0 -> -5, pnonc < 0 -> -6
  // (NaN compares false, so NaN inputs fall through to cumchn exactly as in the reference)
  if (x < 0.0) return 1;
  if (l <= 0.0) return 1;
  if (delta < 0.0) return 1;
  double p, q;
  cumchn(x, l, delta, &p, &q);
  return q;
}

// MixtureChiSquare::getPvalue  (MixtureChiSquare.cpp:7-29)
double orc_davies_pvalue(const double* lambda, int n, double Q, int* fault_out) {
  if (fault_out) *fault_out = 0;
  if (n == 1) return orc_liu_pvalue(lambda, n, Q);
  static thread_local double nc[4096];
  static thread_local int df[4096];
  if (n > 4096) return -1.0;
  for (int i = 0; i < n; ++i) {
    nc[i] = 0.0;
    df[i] = 1;
  }
  int fault;
  double trace[7];
  double pValue = 1.0 - orc_qf(lambda, nc, df, n, 0.0, Q, 10000, 0.000001, trace, &fault);
  if (pValue > 1.0) pValue = 1.0;
  if (fault) pValue = -1.0;
  if (fault_out) *fault_out = fault;
  return pValue;
}

}  // extern "C"
