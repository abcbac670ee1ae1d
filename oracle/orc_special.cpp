// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
//
// CPU restatement of the scalar special functions the rvtests kernel/burden hot path
// calls through GSL 1.16 (the reference vendors third/gsl-1.16.tar.gz):
//   gsl_ran_beta_pdf      randist/beta.c:43-75      (used by src/Model.h:2651-2655, 2806-2810)
//   gsl_ran_chisq_pdf     randist/chisq.c:38-60     (regression/SkatO.cpp:325,336)
//   gsl_cdf_chisq_P/Q     cdf/chisq.c:24-34 -> cdf/gamma.c:28-74 -> specfunc/gamma_inc.c
//                         (LinearRegressionScoreTest.cpp:259-261, SkatO.cpp:336,421)
//   gsl_cdf_chisq_Qinv    cdf/chisqinv.c:31-34 -> cdf/gammainv.c:118-190 (SkatO.cpp:430)
// The regularised incomplete gamma follows the published algorithms GSL uses (power series for
// P, modified-Lentz continued fraction for Q, large-x asymptotic); lgamma comes from libm.
// Pinned against GSL 1.16 itself through tests/golden/gsl_scalar.json (generated in the build
// container by tests/golden/make_gsl_golden.py from the vendored tarball).
#include <cfloat>
#include <cmath>
#include "orc_api.h"

namespace {

const double kEps = DBL_EPSILON;

// log(1+x) - x, accurate for small |x|  (GSL gsl_sf_log_1plusx_mx)
double log1pmx(double x) {
  if (std::fabs(x) < 0.5) {
    // series: -x^2/2 + x^3/3 - x^4/4 + ...
    double term = x, sum = 0.0;
    for (int n = 2; n < 200; ++n) {
      term *= -x;                 // (-1)^{n-1} x^n
      const double t = term / n;
      sum += t;
      if (std::fabs(t) < 1e-18 * std::fabs(sum)) break;
    }
    return sum;
  }
  return std::log1p(x) - x;
}

// gammastar(a) = Gamma(a) / (sqrt(2 pi) a^{a-1/2} e^{-a}),  a >= 10 : Stirling series
double gammastar_large(double a) {
  const double y = 1.0 / (a * a);
  // ln gammastar = 1/(12a) - 1/(360a^3) + 1/(1260a^5) - 1/(1680a^7) + 1/(1188 a^9)
  const double ser = (1.0 / 12.0 +
                      y * (-1.0 / 360.0 + y * (1.0 / 1260.0 + y * (-1.0 / 1680.0 + y * (1.0 / 1188.0))))) /
                     a;
  return std::exp(ser);
}

// D(a,x) = x^a e^{-x} / Gamma(a+1)      (specfunc/gamma_inc.c:37-77)
double gamma_inc_D(double a, double x) {
  if (a < 10.0) {
    const double lnr = a * std::log(x) - x - std::lgamma(a + 1.0);
    return std::exp(lnr);
  }
  double ln_term;
  if (x < 0.5 * a) {
    const double u = x / a;
    ln_term = std::log(u) - u + 1.0;
  } else {
    const double mu = (x - a) / a;
    ln_term = log1pmx(mu);
  }
  const double term1 = std::exp(a * ln_term) / std::sqrt(2.0 * M_PI * a);
  return term1 / gammastar_large(a);
}

// P series   (specfunc/gamma_inc.c:82-150)
double gamma_inc_P_series(double a, double x) {
  const int nmax = 10000;
  const double D = gamma_inc_D(a, x);
  double sum = 1.0, term = 1.0;
  int n;
  const int nlow = (x > a) ? (int)(x - a) : 0;
  for (n = 1; n < nlow; n++) {
    term *= x / (a + n);
    sum += term;
  }
  for (; n < nmax; n++) {
    term *= x / (a + n);
    sum += term;
    if (std::fabs(term / sum) < kEps) break;
  }
  return D * sum;
}

// continued fraction F(a,x), modified Lentz   (specfunc/gamma_inc.c:246-290)
double gamma_inc_F_CF(double a, double x) {
  const int nmax = 5000;
  const double small = kEps * kEps * kEps;
  double hn = 1.0, Cn = 1.0 / small, Dn = 1.0;
  for (int n = 2; n < nmax; n++) {
    const double an = (n & 1) ? 0.5 * (n - 1) / x : (0.5 * n - a) / x;
    Dn = 1.0 + an * Dn;
    if (std::fabs(Dn) < small) Dn = small;
    Cn = 1.0 + an / Cn;
    if (std::fabs(Cn) < small) Cn = small;
    Dn = 1.0 / Dn;
    const double delta = Cn * Dn;
    hn *= delta;
    if (std::fabs(delta - 1.0) < kEps) break;
  }
  return hn;
}

double gamma_inc_Q_CF(double a, double x) { return gamma_inc_D(a, x) * (a / x) * gamma_inc_F_CF(a, x); }

double gamma_inc_Q_large_x(double a, double x) {
  const int nmax = 5000;
  const double D = gamma_inc_D(a, x);
  double sum = 1.0, term = 1.0, last = 1.0;
  for (int n = 1; n < nmax; n++) {
    term *= (a - n) / x;
    if (std::fabs(term / last) > 1.0) break;
    if (std::fabs(term / sum) < kEps) break;
    sum += term;
    last = term;
  }
  return D * (a / x) * sum;
}

// inverse of the standard normal upper tail, used only as a starting guess (AS241-class
// rational approximation by P. J. Acklam; 1e-9 relative is ample for a Newton start).
double ugaussian_Qinv_guess(double Q) {
  static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                             1.383577518672690e+02,  -3.066479806614716e+01, 2.506628277459239e+00};
  static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                             6.680131188771972e+01, -1.328068155288572e+01};
  static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                             -2.549732539343734e+00, 4.374664141464968e+00,  2.938163982698783e+00};
  static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
                             3.754408661907416e+00};
  const double p = 1.0 - Q;  // lower tail
  double x;
  if (p < 0.02425) {
    const double q = std::sqrt(-2 * std::log(p));
    x = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
        ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
  } else if (p <= 1 - 0.02425) {
    const double q = p - 0.5, r = q * q;
    x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
        (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1);
  } else {
    const double q = std::sqrt(-2 * std::log(1 - p));
    x = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
        ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
  }
  return x;
}

}  // namespace

extern "C" {

// gsl_sf_gamma_inc_Q   (specfunc/gamma_inc.c:499-577); the a<0.2 analytic-cancellation series and
// the a>=1e6 uniform asymptotic are not reachable from the hot path (a = df/2 >= 0.5, a << 1e6)
// and fall through to the series / continued fraction here.
double orc_gamma_inc_Q(double a, double x) {
  if (a < 0.0 || x < 0.0) return NAN;
  if (x == 0.0) return 1.0;
  if (a == 0.0) return 0.0;
  if (x <= 0.5 * a) return 1.0 - gamma_inc_P_series(a, x);
  if (a <= x) {
    if (x <= 1.0e6) return gamma_inc_Q_CF(a, x);
    return gamma_inc_Q_large_x(a, x);
  }
  if (x > a - std::sqrt(a)) return gamma_inc_Q_CF(a, x);
  return 1.0 - gamma_inc_P_series(a, x);
}

// gsl_sf_gamma_inc_P   (specfunc/gamma_inc.c:580-642)
double orc_gamma_inc_P(double a, double x) {
  if (a <= 0.0 || x < 0.0) return NAN;
  if (x == 0.0) return 0.0;
  if (x < 20.0 || x < 0.5 * a) return gamma_inc_P_series(a, x);
  if (a <= x) {
    const double Q = (a > 0.2 * x) ? gamma_inc_Q_CF(a, x) : gamma_inc_Q_large_x(a, x);
    return 1.0 - Q;
  }
  if ((x - a) * (x - a) < a) return 1.0 - gamma_inc_Q_CF(a, x);
  return gamma_inc_P_series(a, x);
}

// gsl_cdf_gamma_P / _Q with scale b   (cdf/gamma.c:28-74)
double orc_gamma_cdf_P(double x, double a, double b) {
  if (x <= 0.0) return 0.0;
  const double y = x / b;
  return (y > a) ? 1.0 - orc_gamma_inc_Q(a, y) : orc_gamma_inc_P(a, y);
}
double orc_gamma_cdf_Q(double x, double a, double b) {
  if (x <= 0.0) return 1.0;
  const double y = x / b;
  return (y < a) ? 1.0 - orc_gamma_inc_P(a, y) : orc_gamma_inc_Q(a, y);
}
double orc_chisq_P(double x, double nu) { return orc_gamma_cdf_P(x, nu / 2, 2.0); }
double orc_chisq_Q(double x, double nu) { return orc_gamma_cdf_Q(x, nu / 2, 2.0); }

// gsl_ran_gamma_pdf  (randist/gamma.c:151-175)
double orc_gamma_pdf(double x, double a, double b) {
  if (x < 0) return 0;
  if (x == 0) return (a == 1) ? 1 / b : 0;
  if (a == 1) return std::exp(-x / b) / b;
  return std::exp((a - 1) * std::log(x / b) - x / b - std::lgamma(a)) / b;
}

// gsl_ran_chisq_pdf  (randist/chisq.c:38-60)
double orc_chisq_pdf(double x, double nu) {
  if (x < 0) return 0;
  if (nu == 2.0) return std::exp(-x / 2.0) / 2.0;
  return std::exp((nu / 2 - 1) * std::log(x / 2) - x / 2 - std::lgamma(nu / 2)) / 2;
}

// gsl_ran_beta_pdf  (randist/beta.c:43-75)
double orc_beta_pdf(double x, double a, double b) {
  if (x < 0 || x > 1) return 0;
  const double gab = std::lgamma(a + b), ga = std::lgamma(a), gb = std::lgamma(b);
  if (x == 0.0 || x == 1.0) {
    if (a > 1.0 && b > 1.0) return 0.0;
    return std::exp(gab - ga - gb) * std::pow(x, a - 1) * std::pow(1 - x, b - 1);
  }
  return std::exp(gab - ga - gb + std::log(x) * (a - 1) + std::log1p(-x) * (b - 1));
}

// gsl_cdf_gamma_Qinv  (cdf/gammainv.c:118-190)  — same start values and step rule.
double orc_gamma_cdf_Qinv(double Q, double a, double b) {
  if (Q == 1.0) return 0.0;
  if (Q == 0.0) return INFINITY;
  double x;
  if (Q < 0.05) {
    x = -std::log(Q) + std::lgamma(a);
  } else if (Q > 0.95) {
    x = std::exp((std::lgamma(a) + std::log1p(-Q)) / a);
  } else {
    const double xg = ugaussian_Qinv_guess(Q);
    x = (xg < -0.5 * std::sqrt(a)) ? a : std::sqrt(a) * xg + a;
  }
  unsigned n = 0;
  for (;;) {
    const double dQ = Q - orc_gamma_cdf_Q(x, a, 1.0);
    const double phi = orc_gamma_pdf(x, a, 1.0);
    if (dQ == 0.0 || n++ > 32) break;
    const double lambda = -dQ / std::fmax(2 * std::fabs(dQ / x), phi);
    const double step0 = lambda;
    const double step1 = -((a - 1) / x - 1) * lambda * lambda / 4.0;
    double step = step0;
    if (std::fabs(step1) < 0.5 * std::fabs(step0)) step += step1;
    if (x + step > 0)
      x += step;
    else
      x /= 2.0;
    if (!(std::fabs(step0) > 1e-10 * x)) break;
  }
  return b * x;
}
double orc_chisq_Qinv(double Q, double nu) { return orc_gamma_cdf_Qinv(Q, nu / 2, 2.0); }

}  // extern "C"
