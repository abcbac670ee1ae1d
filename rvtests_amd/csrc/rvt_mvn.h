// rvtests_amd — multivariate-normal band probability  P(-T < Z_i < T, i = 1..n),  Z ~ N(0, R)  (device + host).
//
// AnalyticVT's p-value (src/Model.h:2105-2259 -> MultivariateVT::compute, regression/MultivariateVT.cpp:22-144 ->
// MvtNorm::compute_Band, regression/libMvtnorm/mvtnorm.cpp:57-72) is 1 - this probability for the correlation matrix of
// the nested threshold statistics.  The reference evaluates it with Genz's MVTDST (regression/libMvtnorm/mvt.f): a
// RANDOMISED quasi-Monte-Carlo rule driven by rand(), at most 25 000 points, requested absolute accuracy 1e-3 — its
// digits beyond ~1e-3 are noise and differ from run to run of the process-wide random stream.  There is therefore no
// 1e-6 parity to be had; what is computed here is the SAME integral by the same separation-of-variables transformation
// (Genz 1992: Cholesky factor, conditional limits, one uniform per dimension), evaluated deterministically on a fixed
// lattice with far more points and with Genz's variable reordering, so that the result is accurate to ~1e-5 and reproducible:
//   e_1 = Phi(T / l_11), d_1 = Phi(-T / l_11), f = e_1 - d_1
//   for i = 2..n:  y_{i-1} = Phi^-1(d_{i-1} + w_{i-1} (e_{i-1} - d_{i-1})),  s = sum_{j<i} l_ij y_j,
//                  d_i = Phi((-T - s) / l_ii), e_i = Phi((T - s) / l_ii),  f *= e_i - d_i
//   (a zero pivot l_ii — a threshold statistic that is a linear combination of earlier ones — contributes the
//   indicator of -T < s < T).
// Points: rank-1 lattice w_i = |2 frac(k alpha_i + shift_i) - 1| with alpha_i = frac(sqrt(p_i)), p_i the i-th prime
// (Richtmyer), kMvnShifts shifts from a fixed linear congruential sequence; the spread of the shift means is the error
// estimate.
#pragma once
#include <math.h>
#include <stdint.h>
#include "rvt_special.h"

namespace rvt {

constexpr int kMvnMaxDim = 256;     // threshold statistics per gene handled on the device
constexpr int kMvnShifts = 8;       // independent lattice shifts (error estimate)
constexpr int kMvnPoints = 16384;   // lattice points per shift at most (doubled from 1024 until the error estimate is small)

RVT_HDI double mvn_phi(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }

// inverse normal cdf: P. J. Acklam's rational approximation (relative error ~1e-9) refined by one Halley step on Phi
RVT_HDI double mvn_phiinv(double p) {
  if (!(p > 0.0)) return -INFINITY;
  if (!(p < 1.0)) return INFINITY;
  const double plow = 0.02425;
  double x;
  if (p < plow || p > 1.0 - plow) {
    const double pp = p < plow ? p : 1.0 - p;
    const double q = sqrt(-2.0 * log(pp));
    x = (((((-7.784894002430293e-03 * q - 3.223964580411365e-01) * q - 2.400758277161838e+00) * q - 2.549732539343734e+00) * q +
          4.374664141464968e+00) * q + 2.938163982698783e+00) /
        ((((7.784695709041462e-03 * q + 3.224671290700398e-01) * q + 2.445134137142996e+00) * q + 3.754408661907416e+00) * q + 1.0);
    if (p > 1.0 - plow) x = -x;
  } else {
    const double q = p - 0.5, r = q * q;
    x = (((((-3.969683028665376e+01 * r + 2.209460984245205e+02) * r - 2.759285104469687e+02) * r + 1.383577518672690e+02) * r -
          3.066479806614716e+01) * r + 2.506628277459239e+00) * q /
        (((((-5.447609879822406e+01 * r + 1.615858368580409e+02) * r - 1.556989798598866e+02) * r + 6.680131188771972e+01) * r -
          1.328068155288572e+01) * r + 1.0);
  }
  // one Halley step on the tail that keeps relative accuracy: err = Phi(x) - p evaluated through the nearer tail
  const bool upper = x > 0.0;
  const double tail = upper ? mvn_phi(-x) : mvn_phi(x);        // Phi of the nearer tail
  const double target = upper ? 1.0 - p : p;
  const double pdf = 0.39894228040143267794 * exp(-0.5 * x * x);
  if (pdf > 0.0) {
    const double err = upper ? target - tail : tail - target;  // Phi(x) - p
    const double u = err / pdf;
    x -= u / (1.0 + 0.5 * x * u);
  }
  return x;
}

RVT_HDI double mvn_pdf(double x) { return 0.39894228040143267794 * exp(-0.5 * x * x); }

// Expected width of the conditional integration interval of a candidate variable with conditional variance v and
// conditional mean shift s (Genz's ordering criterion); *ex receives the conditional expectation of its standardised
// value over that interval.
RVT_HDI double mvn_candidate_width(double v, double s, double T, double tol, double* ex) {
  if (v > tol) {
    const double c = sqrt(v), a = (-T - s) / c, b = (T - s) / c;
    const double w = mvn_phi(b) - mvn_phi(a);
    *ex = w > 0.0 ? (mvn_pdf(a) - mvn_pdf(b)) / w : 0.0;
    return w;
  }
  *ex = 0.0;
  return (s > -T && s < T) ? 1.0 : 0.0;
}

// In-place lower Cholesky factor of the symmetric n x n correlation matrix A (row-major, leading dimension lda; the
// FULL matrix must be stored) with Genz's variable reordering for the band (-T, T)^n: at step i the remaining variable
// with the smallest expected conditional interval width comes next (rows and columns are swapped; all limits are equal,
// so nothing else moves), which concentrates the variation of the integrand in the first dimensions.  Pivots below
// `tol` become exact zeros with a zero column below them.  yexp: n doubles of scratch.  Returns the number of zero pivots.
RVT_HDI int mvn_cholesky(double* A, int n, int lda, double tol, double T, double* yexp) {
  int zeros = 0;
  for (int i = 0; i < n; ++i) {
    int best = i;
    double bestw = 2.0, bestv = 0.0, bests = 0.0, bestex = 0.0;
    for (int j = i; j < n; ++j) {
      double v = A[j * lda + j], sh = 0.0;
      for (int k = 0; k < i; ++k) {
        v -= A[j * lda + k] * A[j * lda + k];
        sh += A[j * lda + k] * yexp[k];
      }
      double ex;
      const double w = mvn_candidate_width(v, sh, T, tol, &ex);
      if (w < bestw) {
        bestw = w;
        best = j;
        bestv = v;
        bests = sh;
        bestex = ex;
      }
    }
    (void)bests;
    if (best != i) {
      for (int k = 0; k < n; ++k) {  // rows
        const double t = A[i * lda + k];
        A[i * lda + k] = A[best * lda + k];
        A[best * lda + k] = t;
      }
      for (int r = 0; r < n; ++r) {  // columns
        const double t = A[r * lda + i];
        A[r * lda + i] = A[r * lda + best];
        A[r * lda + best] = t;
      }
    }
    yexp[i] = bestex;
    if (bestv > tol) {
      const double l = sqrt(bestv);
      A[i * lda + i] = l;
      for (int r = i + 1; r < n; ++r) {
        double t = A[r * lda + i];
        for (int k = 0; k < i; ++k) t -= A[r * lda + k] * A[i * lda + k];
        A[r * lda + i] = t / l;
      }
    } else {
      ++zeros;
      A[i * lda + i] = 0.0;
      for (int r = i + 1; r < n; ++r) A[r * lda + i] = 0.0;
    }
  }
  return zeros;
}

// the i-th lattice generator frac(sqrt(prime_i)); primes[] holds the first n primes
RVT_HDI double mvn_alpha(int prime) {
  const double r = sqrt((double)prime);
  return r - floor(r);
}

// shift j (0 .. kMvnShifts-1), dimension i: a fixed 64-bit mixing sequence -> [0, 1)
RVT_HDI double mvn_shift(int j, int i) {
  uint64_t z = 0x9E3779B97F4A7C15ull * (uint64_t)(j * 4099 + i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// integrand at lattice point k of shift j.  L: lower factor (row-major, lda), y: n doubles of scratch, element q at
// y[q * ys] (the device keeps one element of all 256 threads side by side)
RVT_HDI double mvn_band_point_strided(const double* L, int n, int lda, double T, const double* alpha, int j, long long k,
                                      double* y, int ys) {
  double f = 1.0;
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int q = 0; q < i; ++q) s += L[i * lda + q] * y[(size_t)q * ys];
    const double l = L[i * lda + i];
    if (l > 0.0) {
      const double dlo = mvn_phi((-T - s) / l), dhi = mvn_phi((T - s) / l);
      const double w = dhi - dlo;
      f *= w;
      if (!(f > 0.0)) return 0.0;
      if (i + 1 < n) {
        double x = (double)k * alpha[i] + mvn_shift(j, i);
        x -= floor(x);
        x = fabs(2.0 * x - 1.0);
        // keep the uniform strictly inside (0, 1): Phi^-1 of an end point is infinite
        x = fmin(fmax(x, 1e-15), 1.0 - 1e-15);
        y[(size_t)i * ys] = mvn_phiinv(dlo + x * w);
      }
    } else {
      if (!(s > -T && s < T)) return 0.0;
      y[(size_t)i * ys] = 0.0;  // (its column of L is zero: the value is never used)
    }
  }
  return f;
}
RVT_HDI double mvn_band_point(const double* L, int n, int lda, double T, const double* alpha, int j, long long k,
                              double* y) {
  return mvn_band_point_strided(L, n, lda, T, alpha, j, k, y, 1);
}

// Whole integral on one host thread (hostcheck / tests): A = correlation matrix (row-major n x n, destroyed).
// Returns the probability; *err = 3.5 standard errors over the shifts.
RVT_HD double mvn_band_prob_serial(double* A, int n, double T, double* y, double* alpha, double* err) {
  if (n == 1) {
    if (err) *err = 0.0;
    return mvn_phi(T) - mvn_phi(-T);
  }
  mvn_cholesky(A, n, n, 1e-10, T, y);
  int found = 0;
  for (int cand = 2; found < n; ++cand) {
    bool prime = true;
    for (int q = 2; q * q <= cand; ++q)
      if (cand % q == 0) {
        prime = false;
        break;
      }
    if (prime) alpha[found++] = mvn_alpha(cand);
  }
  double acc[kMvnShifts];
  for (int j = 0; j < kMvnShifts; ++j) acc[j] = 0.0;
  long long done = 0;
  double est = 0.0, e = 1.0;
  for (long long P = 1024; P <= kMvnPoints; P *= 2) {
    double mean = 0.0, sq = 0.0;
    for (int j = 0; j < kMvnShifts; ++j) {
      for (long long k = done; k < P; ++k) acc[j] += mvn_band_point(A, n, n, T, alpha, j, k + 1, y);
      const double mj = acc[j] / (double)P;
      mean += mj;
      sq += mj * mj;
    }
    done = P;
    mean /= kMvnShifts;
    const double var = fmax(0.0, sq / kMvnShifts - mean * mean) / (kMvnShifts - 1);
    est = mean;
    e = 3.5 * sqrt(var);
    if (e < 5e-5) break;
  }
  if (err) *err = e;
  return est;
}

}  // namespace rvt
