// rvtests_amd — multivariate-normal band probability  P(-T < Z_i < T, i = 1..n),  Z ~ N(0, R)  (device + host).
//
// AnalyticVT's p-value (src/Model.h:2105-2259 -> MultivariateVT::compute, regression/MultivariateVT.cpp:22-144 ->
// MvtNorm::compute_Band, regression/libMvtnorm/mvtnorm.cpp:57-72) is 1 - this probability for the correlation matrix of
// the nested threshold statistics.  The reference evaluates it with Genz's MVTDST (regression/libMvtnorm/mvt.f): a
// RANDOMISED quasi-Monte-Carlo rule driven by rand(), at most 25 000 points, requested absolute accuracy 1e-3 — its
// digits beyond ~1e-3 are noise and differ from run to run of the process-wide random stream.  There is therefore no
// 1e-6 parity to be had; what is computed here is the SAME integral by the same separation-of-variables transformation
// (Genz 1992: Cholesky factor, conditional limits, one uniform per dimension), evaluated deterministically on a fixed
// lattice with far more points, so that the result is accurate to ~1e-5 and reproducible:
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

// inverse normal cdf: rational starting value (Abramowitz & Stegun 26.2.23), then Halley steps on Phi
RVT_HDI double mvn_phiinv(double p) {
  if (!(p > 0.0)) return -INFINITY;
  if (!(p < 1.0)) return INFINITY;
  const bool upper = p > 0.5;
  const double q = upper ? 1.0 - p : p;
  const double t = sqrt(-2.0 * log(q));
  double x = t - (2.515517 + t * (0.802853 + t * 0.010328)) / (1.0 + t * (1.432788 + t * (0.189269 + t * 0.001308)));
  x = -x;  // lower-tail quantile of q
  for (int it = 0; it < 3; ++it) {
    const double err = mvn_phi(x) - q;
    const double pdf = 0.39894228040143267794 * exp(-0.5 * x * x);
    if (!(pdf > 0.0)) break;
    const double u = err / pdf;
    x -= u / (1.0 + 0.5 * x * u);
  }
  return upper ? -x : x;
}

// In-place lower Cholesky factor of the symmetric n x n matrix A (row-major, leading dimension lda); pivots below
// `tol` become exact zeros with a zero column below them.  Returns the number of zero pivots.
RVT_HDI int mvn_cholesky(double* A, int n, int lda, double tol) {
  int zeros = 0;
  for (int j = 0; j < n; ++j) {
    double s = A[j * lda + j];
    for (int k = 0; k < j; ++k) s -= A[j * lda + k] * A[j * lda + k];
    if (s > tol) {
      const double l = sqrt(s);
      A[j * lda + j] = l;
      for (int i = j + 1; i < n; ++i) {
        double t = A[i * lda + j];
        for (int k = 0; k < j; ++k) t -= A[i * lda + k] * A[j * lda + k];
        A[i * lda + j] = t / l;
      }
    } else {
      ++zeros;
      A[j * lda + j] = 0.0;
      for (int i = j + 1; i < n; ++i) A[i * lda + j] = 0.0;
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
  mvn_cholesky(A, n, n, 1e-10);
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
