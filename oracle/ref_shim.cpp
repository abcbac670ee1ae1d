// ORACLE/_ref — TEST INFRASTRUCTURE ONLY.
// Thin extern "C" shim over the REAL reference sources, compiled where they lie under
// /root/reference (never copied):  regression/MixtureChiSquare.cpp (which #includes qfc.c) and
// regression/cdflib.cpp.  These three files have no third-party dependency, so they build with
// plain g++.  Output goes to oracle/_ref/libref_mixchisq.so (git-ignored).
// Used to validate the restatement in oracle/orc_davies.cpp / orc_liu.cpp / orc_special.cpp.
#include "MixtureChiSquare.h"
#include "cdflib.h"

// defined (non-static) in qfc.c, which MixtureChiSquare.cpp includes
double qf(double* lb1, double* nc1, int* n1, int r1, double sigma, double c1, int lim1, double acc,
          double* trace, int* ifault);

extern "C" {

double ref_davies_pvalue(const double* lambda, int n, double Q) {
  MixtureChiSquare m;
  for (int i = 0; i < n; ++i) m.addLambda(lambda[i]);
  return m.getPvalue(Q);
}

double ref_liu_pvalue(const double* lambda, int n, double Q) {
  MixtureChiSquare m;
  for (int i = 0; i < n; ++i) m.addLambda(lambda[i]);
  return m.getLiuPvalue(Q);
}

double ref_qf(double* lb, double* nc, int* n, int r, double sigma, double c, int lim, double acc, double* trace,
              int* ifault) {
  return qf(lb, nc, n, r, sigma, c, lim, acc, trace, ifault);
}

void ref_cumchn(double x, double df, double pnonc, double* cum, double* ccum) {
  cumchn(&x, &df, &pnonc, cum, ccum);
}

void ref_cumchi(double x, double df, double* cum, double* ccum) { cumchi(&x, &df, cum, ccum); }

void ref_gamma_inc(double a, double x, double* ans, double* qans) {
  int ind = 0;
  gamma_inc(&a, &x, ans, qans, &ind);
}

}  // extern "C"
