// rvtests_amd — scalar special functions used by the per-gene p-value kernels (device code).
//
// Every function is `RVT_HD` so the SAME source is compiled by hipcc for gfx950 (the product path,
// kernels in pvalue_kernels.hip) and by g++ into the test-only host harness
// (csrc/hostcheck.cpp -> librvt_hostcheck.so) that lets the CPU test-suite exercise the device
// algorithms without a GPU.  The shipped library never calls the host instantiations.
//
// Replaces, for the hot path, the GSL 1.16 entry points the reference links:
//   gsl_ran_beta_pdf   (reference call sites src/Model.h:2651-2655, 2806-2810)
//   gsl_cdf_chisq_Q/P  (regression/LinearRegressionScoreTest.cpp:259-261, SkatO.cpp:336,421)
//   gsl_cdf_chisq_Qinv (regression/SkatO.cpp:430)
//   gsl_ran_chisq_pdf  (regression/SkatO.cpp:325,336)
#pragma once
#include <math.h>
#include <float.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RVT_HD __host__ __device__ inline
#else
#define RVT_HD inline
#endif

namespace rvt {

constexpr double kDblEps = 2.2204460492503131e-16;
constexpr double kPi = 3.14159265358979323846;

// log(1+x) - x without cancellation
RVT_HD double log1p_minus_x(double x) {
  if (fabs(x) >= 0.5) return log1p(x) - x;
  // -x²/2 + x³/3 - ...   (|x| < 0.5: < 60 terms to 1e-18)
  double p = x, acc = 0.0;
  for (int n = 2; n < 128; ++n) {
    p *= -x;
    const double t = p / (double)n;
    acc += t;
    if (fabs(t) < 1e-18 * fabs(acc)) break;
  }
  return acc;
}

// prefactor x^a e^-x / Gamma(a+1), evaluated in a cancellation-free form for large a
RVT_HD double igam_prefactor(double a, double x) {
  if (a < 10.0) return exp(a * log(x) - x - lgamma(a + 1.0));
  double lt;
  if (x < 0.5 * a) {
    const double u = x / a;
    lt = log(u) - u + 1.0;
  } else {
    lt = log1p_minus_x((x - a) / a);
  }
  const double y = 1.0 / (a * a);
  // Stirling: ln Gamma*(a) = 1/(12a) - 1/(360a³) + 1/(1260a⁵) - 1/(1680a⁷) + 1/(1188a⁹)
  const double lgs =
      (1.0 / 12.0 + y * (-1.0 / 360.0 + y * (1.0 / 1260.0 + y * (-1.0 / 1680.0 + y * (1.0 / 1188.0))))) / a;
  return exp(a * lt - lgs) / sqrt(2.0 * kPi * a);
}

// lower regularised incomplete gamma by its power series
RVT_HD double igam_series_P(double a, double x) {
  const double pre = igam_prefactor(a, x);
  double term = 1.0, sum = 1.0;
  int n = 1;
  const int nrise = (x > a) ? (int)(x - a) : 0;  // terms still growing
  for (; n < nrise; ++n) {
    term *= x / (a + n);
    sum += term;
  }
  for (; n < 10000; ++n) {
    term *= x / (a + n);
    sum += term;
    if (fabs(term / sum) < kDblEps) break;
  }
  return pre * sum;
}

// upper regularised incomplete gamma by the Legendre continued fraction (modified Lentz)
RVT_HD double igam_cf_Q(double a, double x) {
  const double tiny = kDblEps * kDblEps * kDblEps;
  double h = 1.0, Cn = 1.0 / tiny, Dn = 1.0;
  for (int n = 2; n < 5000; ++n) {
    const double an = (n & 1) ? 0.5 * (double)(n - 1) / x : (0.5 * (double)n - a) / x;
    Dn = 1.0 + an * Dn;
    if (fabs(Dn) < tiny) Dn = tiny;
    Cn = 1.0 + an / Cn;
    if (fabs(Cn) < tiny) Cn = tiny;
    Dn = 1.0 / Dn;
    const double del = Cn * Dn;
    h *= del;
    if (fabs(del - 1.0) < kDblEps) break;
  }
  return igam_prefactor(a, x) * (a / x) * h;
}

RVT_HD double igam_asym_Q(double a, double x) {  // x >> a
  double sum = 1.0, term = 1.0, last = 1.0;
  for (int n = 1; n < 5000; ++n) {
    term *= (a - n) / x;
    if (fabs(term / last) > 1.0) break;
    if (fabs(term / sum) < kDblEps) break;
    sum += term;
    last = term;
  }
  return igam_prefactor(a, x) * (a / x) * sum;
}

RVT_HD double igam_Q(double a, double x) {
  if (a < 0.0 || x < 0.0) return NAN;
  if (x == 0.0) return 1.0;
  if (a == 0.0) return 0.0;
  if (x <= 0.5 * a) return 1.0 - igam_series_P(a, x);
  if (a <= x) return (x <= 1.0e6) ? igam_cf_Q(a, x) : igam_asym_Q(a, x);
  if (x > a - sqrt(a)) return igam_cf_Q(a, x);
  return 1.0 - igam_series_P(a, x);
}

RVT_HD double igam_P(double a, double x) {
  if (a <= 0.0 || x < 0.0) return NAN;
  if (x == 0.0) return 0.0;
  if (x < 20.0 || x < 0.5 * a) return igam_series_P(a, x);
  if (a <= x) return 1.0 - ((a > 0.2 * x) ? igam_cf_Q(a, x) : igam_asym_Q(a, x));
  if ((x - a) * (x - a) < a) return 1.0 - igam_cf_Q(a, x);
  return igam_series_P(a, x);
}

// gamma distribution tails with scale b (the form gsl_cdf_gamma_{P,Q} takes)
RVT_HD double gamma_tail_Q(double x, double a, double b) {
  if (x <= 0.0) return 1.0;
  const double y = x / b;
  return (y < a) ? 1.0 - igam_P(a, y) : igam_Q(a, y);
}
RVT_HD double gamma_tail_P(double x, double a, double b) {
  if (x <= 0.0) return 0.0;
  const double y = x / b;
  return (y > a) ? 1.0 - igam_Q(a, y) : igam_P(a, y);
}
RVT_HD double chisq_Q(double x, double nu) { return gamma_tail_Q(x, 0.5 * nu, 2.0); }
RVT_HD double chisq_P(double x, double nu) { return gamma_tail_P(x, 0.5 * nu, 2.0); }

RVT_HD double gamma_density(double x, double a, double b) {
  if (x < 0) return 0.0;
  if (x == 0) return (a == 1) ? 1.0 / b : 0.0;
  if (a == 1) return exp(-x / b) / b;
  return exp((a - 1) * log(x / b) - x / b - lgamma(a)) / b;
}

// lg = lgamma(nu / 2), passed in so that a caller evaluating many x for one nu computes it once
RVT_HD double chisq_density_lg(double x, double nu, double lg) {
  if (x < 0) return 0.0;
  if (nu == 2.0) return exp(-x / 2.0) / 2.0;
  return exp((nu / 2 - 1) * log(x / 2) - x / 2 - lg) / 2;
}
RVT_HD double chisq_density(double x, double nu) { return chisq_density_lg(x, nu, lgamma(nu / 2)); }

RVT_HD double beta_density(double x, double a, double b) {
  if (x < 0 || x > 1) return 0.0;
  const double lnorm = lgamma(a + b) - lgamma(a) - lgamma(b);
  if (x == 0.0 || x == 1.0) {
    if (a > 1.0 && b > 1.0) return 0.0;
    return exp(lnorm) * pow(x, a - 1) * pow(1 - x, b - 1);
  }
  return exp(lnorm + log(x) * (a - 1) + log1p(-x) * (b - 1));
}

// starting point for the normal quantile (Acklam's rational approximation, ~1e-9)
RVT_HD double normal_upper_quantile_guess(double Q) {
  const double p = 1.0 - Q;
  const double a0 = -3.969683028665376e+01, a1 = 2.209460984245205e+02, a2 = -2.759285104469687e+02,
               a3 = 1.383577518672690e+02, a4 = -3.066479806614716e+01, a5 = 2.506628277459239e+00;
  const double b0 = -5.447609879822406e+01, b1 = 1.615858368580409e+02, b2 = -1.556989798598866e+02,
               b3 = 6.680131188771972e+01, b4 = -1.328068155288572e+01;
  const double c0 = -7.784894002430293e-03, c1 = -3.223964580411365e-01, c2 = -2.400758277161838e+00,
               c3 = -2.549732539343734e+00, c4 = 4.374664141464968e+00, c5 = 2.938163982698783e+00;
  const double d0 = 7.784695709041462e-03, d1 = 3.224671290700398e-01, d2 = 2.445134137142996e+00,
               d3 = 3.754408661907416e+00;
  if (p < 0.02425) {
    const double q = sqrt(-2 * log(p));
    return (((((c0 * q + c1) * q + c2) * q + c3) * q + c4) * q + c5) / ((((d0 * q + d1) * q + d2) * q + d3) * q + 1);
  }
  if (p <= 1 - 0.02425) {
    const double q = p - 0.5, r = q * q;
    return (((((a0 * r + a1) * r + a2) * r + a3) * r + a4) * r + a5) * q /
           (((((b0 * r + b1) * r + b2) * r + b3) * r + b4) * r + 1);
  }
  const double q = sqrt(-2 * log(1 - p));
  return -(((((c0 * q + c1) * q + c2) * q + c3) * q + c4) * q + c5) / ((((d0 * q + d1) * q + d2) * q + d3) * q + 1);
}

// upper-tail quantile of the gamma distribution; same start values / step rule as the routine the
// reference calls (gsl_cdf_gamma_Qinv), so the iterates — and the 1e-10 stopping point — agree.
RVT_HD double gamma_quantile_Q(double Q, double a, double b) {
  if (Q == 1.0) return 0.0;
  if (Q == 0.0) return INFINITY;
  double x;
  if (Q < 0.05)
    x = -log(Q) + lgamma(a);
  else if (Q > 0.95)
    x = exp((lgamma(a) + log1p(-Q)) / a);
  else {
    const double xg = normal_upper_quantile_guess(Q);
    x = (xg < -0.5 * sqrt(a)) ? a : sqrt(a) * xg + a;
  }
  for (unsigned it = 0;; ++it) {
    const double dQ = Q - gamma_tail_Q(x, a, 1.0);
    const double phi = gamma_density(x, a, 1.0);
    if (dQ == 0.0 || it > 32) break;
    const double lam = -dQ / fmax(2 * fabs(dQ / x), phi);
    const double corr = -((a - 1) / x - 1) * lam * lam / 4.0;
    double step = lam;
    if (fabs(corr) < 0.5 * fabs(lam)) step += corr;
    if (x + step > 0)
      x += step;
    else
      x *= 0.5;
    if (!(fabs(lam) > 1e-10 * x)) break;
  }
  return b * x;
}
RVT_HD double chisq_quantile_Q(double Q, double nu) { return gamma_quantile_Q(Q, 0.5 * nu, 2.0); }

}  // namespace rvt
