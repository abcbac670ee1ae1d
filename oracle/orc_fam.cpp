// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_api.h).  Related-sample path: FastLMM null model and FamSKAT.
//
// Restates, in the reference's operation order and (optionally) its fp32 storage:
//   FastLMM::Impl::FitNullModel   regression/FastLMM.cpp:28-142    (model MLE, test SCORE — what FamSkat constructs)
//   getBetaSigma2 / getSumResidual2 / getLogLikelihood             regression/FastLMM.cpp:283-346
//   Minimizer::minimize           regression/GSLMinimizer.cpp:18-66 over GSL 1.16 min/brent.c + min/fsolver.c
//   FastLMM::Impl::FastGetAF      regression/FastLMM.cpp:402-443
//   FamSkat::FamSkatImpl          regression/FamSkat.cpp:34-138    (LITERAL N x N Sigma / SigmaInv / P0)
// Quirks kept on purpose:
//   * lambda <- |lambda| inside FastLMM (:50) but the raw S in FamSkat's Sigma (FamSkat.cpp:48-54);
//   * when the best grid point is on the boundary the member `delta` keeps the LAST grid value exp(10) (:61-100);
//   * after Brent, delta = x_minimum but beta / sigma2 are those of the LAST function evaluation (:812-817);
//   * P0 = Sigma - X (X' Sigma^-1 X)^-1 X'  (Sigma, not Sigma^-1, FamSkat.cpp:56);
//   * weights beta_pdf(FastGetAF; 1, 25) regardless of the model's beta1/beta2, not squared, applied to G (:78-81,129-137);
//   * Davies only, no Liu fallback (:118).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <vector>

#include "orc_api.h"
#include "orc_linalg.h"

using orc::Mat;

namespace {

struct F32 {
  bool on;
  double operator()(double x) const { return on ? (double)(float)x : x; }
};

// ---- GSL 1.16 Brent minimiser as driven by Minimizer::minimize (epsabs 0.001, epsrel 0, maxIter 100) ------------
// returns 0 and *xmin on success, -1 when gsl_min_fminimizer_set rejects the bracket or f is not finite
int brent_minimize(const std::function<double(double)>& f, double start, double lb, double ub, double* xmin) {
  const double golden = 0.3819660;
  const double sqrt_eps = 1.4901161193847656e-08;  // GSL_SQRT_DBL_EPSILON
  double x_lower = lb, x_upper = ub, x_minimum = start;
  // compute_f_values: lower, upper, minimum (fsolver.c:34-44)
  double f_lower = f(x_lower);
  if (!std::isfinite(f_lower)) return -1;
  double f_upper = f(x_upper);
  if (!std::isfinite(f_upper)) return -1;
  double f_minimum = f(x_minimum);
  if (!std::isfinite(f_minimum)) return -1;
  if (x_lower > x_upper) return -1;
  if (x_minimum >= x_upper || x_minimum <= x_lower) return -1;
  if (f_minimum >= f_lower || f_minimum >= f_upper) return -1;  // "endpoints do not enclose a minimum"
  // brent_init
  double v = x_lower + golden * (x_upper - x_lower), w = v, sd = 0, se = 0;
  double f_v = f(v);
  if (!std::isfinite(f_v)) return -1;
  double f_w = f_v;
  int iter = 0;
  for (;;) {
    ++iter;
    // brent_iterate
    const double x_left = x_lower, x_right = x_upper, z = x_minimum;
    double d = se, e = sd;  // (sic) GSL reads d from state->e and e from state->d
    const double f_z = f_minimum;
    const double w_lower = z - x_left, w_upper = x_right - z;
    const double tolerance = sqrt_eps * std::fabs(z);
    double p = 0, q = 0, r = 0;
    const double midpoint = 0.5 * (x_left + x_right);
    if (std::fabs(e) > tolerance) {
      r = (z - w) * (f_z - f_v);
      q = (z - v) * (f_z - f_w);
      p = (z - v) * q - (z - w) * r;
      q = 2 * (q - r);
      if (q > 0)
        p = -p;
      else
        q = -q;
      r = e;
      e = d;
    }
    double u;
    if (std::fabs(p) < std::fabs(0.5 * q * r) && p < q * w_lower && p < q * w_upper) {
      const double t2 = 2 * tolerance;
      d = p / q;
      u = z + d;
      if ((u - x_left) < t2 || (x_right - u) < t2) d = (z < midpoint) ? tolerance : -tolerance;
    } else {
      e = (z < midpoint) ? x_right - z : -(z - x_left);
      d = golden * e;
    }
    if (std::fabs(d) >= tolerance)
      u = z + d;
    else
      u = z + ((d > 0) ? tolerance : -tolerance);
    se = e;
    sd = d;
    const double f_u = f(u);
    if (!std::isfinite(f_u)) return -1;  // GSL_EBADFUNC -> errorLabel
    if (f_u <= f_z) {
      if (u < z) {
        x_upper = z;
        f_upper = f_z;
      } else {
        x_lower = z;
        f_lower = f_z;
      }
      v = w;
      f_v = f_w;
      w = z;
      f_w = f_z;
      x_minimum = u;
      f_minimum = f_u;
    } else {
      if (u < z) {
        x_lower = u;
        f_lower = f_u;
      } else {
        x_upper = u;
        f_upper = f_u;
      }
      if (f_u <= f_w || w == z) {
        v = w;
        f_v = f_w;
        w = u;
        f_w = f_u;
      } else if (f_u <= f_v || v == z || v == w) {
        v = u;
        f_v = f_u;
      }
    }
    *xmin = x_minimum;
    // gsl_min_test_interval(a, b, 0.001, 0.0)
    if (std::fabs(x_upper - x_lower) < 0.001) return 0;
    if (iter >= 100) return 0;
  }
}

struct Lmm {
  int64_t N;
  int d;
  F32 F;
  Mat ux;                   // U'X
  std::vector<double> uy;   // U'y
  std::vector<double> lam;  // |S|
  std::vector<double> beta;
  double sigma2 = 0, delta = 0;

  void beta_sigma2(double dl) {  // getBetaSigma2, MLE
    Mat A(d, d), b(d, 1);
    for (int a = 0; a < d; ++a) {
      for (int c = 0; c < d; ++c) {
        double s = 0;
        for (int64_t i = 0; i < N; ++i) s = F(s + F(F(ux(i, a) * F(1.0 / std::fabs(F(lam[i] + dl)))) * ux(i, c)));
        A(a, c) = s;
      }
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(F(ux(i, a) * F(1.0 / std::fabs(F(lam[i] + dl)))) * uy[i]));
      b(a, 0) = s;
    }
    Mat x;
    orc::sym_solve(A, b, &x);  // .ldlt().solve()
    beta.assign(d, 0.0);
    for (int a = 0; a < d; ++a) beta[a] = F(x(a, 0));
    // getSumResidual2: sum (uy - ux beta)^2 / (lambda + delta)
    double sr = 0;
    for (int64_t i = 0; i < N; ++i) {
      double p = 0;
      for (int a = 0; a < d; ++a) p = F(p + F(ux(i, a) * beta[a]));
      const double r = F(uy[i] - p);
      sr = F(sr + F(F(r * r) / F(lam[i] + dl)));
    }
    sigma2 = sr / (double)N;
  }
  double loglik(double dl) const {  // getLogLikelihood, MLE
    const double n = (double)N;
    double ret = n * std::log(2.0 * 3.14159265358979323846);
    double sl = 0;
    for (int64_t i = 0; i < N; ++i) sl = F(sl + F(std::log(std::fabs(F(lam[i] + dl)))));
    ret += sl;
    ret += n;
    ret += n * std::log(sigma2);
    return -0.5 * ret;
  }
};

Mat wrapd(const double* p, int64_t r, int64_t c) {
  Mat m(r, c);
  std::memcpy(m.a.data(), p, sizeof(double) * (size_t)r * c);
  return m;
}

}  // namespace

extern "C" {

int orc_brent_builtin(int id, double a, double start, double lb, double ub, double* xmin, int* evals,
                      double* last_x) {
  int n = 0;
  double last = NAN;
  auto f = [&](double x) {
    ++n;
    last = x;
    switch (id) {
      case 0: return (x - a) * (x - a);
      case 1: return std::cosh(x - a);
      case 2: return x * x * x * x - a * x;
      case 3: return a / x + std::log(x);
      default: return std::pow(std::fabs(x - a), 1.5);
    }
  };
  *xmin = start;
  const int rc = brent_minimize(f, start, lb, ub, xmin);
  *evals = n;
  *last_x = last;
  return rc;
}

int orc_fastlmm_null(const double* Xp, const double* y, int64_t N, int d, const double* Up, const double* S,
                     int use_float, orc_fam_null* out) {
  std::memset(out, 0, sizeof(*out));
  F32 F{use_float != 0};
  Lmm m;
  m.N = N;
  m.d = d;
  m.F = F;
  m.ux = Mat(N, d);
  m.uy.assign(N, 0.0);
  m.lam.resize(N);
  for (int64_t i = 0; i < N; ++i) m.lam[i] = std::fabs(F(S[i]));
  // rotate: ux = U'X, uy = U'y
  for (int64_t k = 0; k < N; ++k) {
    const double* uk = Up + (size_t)k * N;  // column k of U
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(F(uk[i]) * F(Xp[(size_t)a * N + i])));
      m.ux(k, a) = s;
    }
    double s = 0;
    for (int64_t i = 0; i < N; ++i) s = F(s + F(F(uk[i]) * F(y[i])));
    m.uy[k] = s;
  }
  int maxIndex = -1;
  double maxLogLik = 0;
  for (int i = 0; i <= 100; ++i) {
    m.delta = std::exp(-10. + i * 0.2);
    m.beta_sigma2(m.delta);
    const double ll = m.loglik(m.delta);
    if (std::isnan(ll)) continue;
    if (maxIndex < 0 || ll > maxLogLik) {
      maxIndex = i;
      maxLogLik = ll;
    }
  }
  out->max_index = maxIndex;
  if (maxIndex == 0 || maxIndex == 100 || maxIndex < 0) {
    // boundary: delta (and beta, sigma2) stay at the LAST grid point
  } else {
    const double lb = std::exp(-10. + (maxIndex - 1) * 0.2), ub = std::exp(-10. + (maxIndex + 1) * 0.2);
    const double start = std::exp(-10. + maxIndex * 0.2);
    int evals = 0;
    auto goal = [&](double x) {
      ++evals;
      m.beta_sigma2(x);
      return -m.loglik(x);
    };
    double xmin = start;
    if (brent_minimize(goal, start, lb, ub, &xmin))
      m.delta = start;
    else
      m.delta = xmin;
    out->brent_evals = evals;
  }
  out->ok = 1;
  out->delta = m.delta;
  out->sigma2 = m.sigma2;
  for (int a = 0; a < d && a < 16; ++a) out->beta[a] = m.beta[a];
  return 0;
}

// FamSkat::FitNullModel + TestCovariate with the LITERAL N x N matrices (small N only).
int orc_famskat(const double* Gp, int64_t N, int M, const double* Xp, const double* y, int d, const double* Up,
                const double* S, const orc_fam_null* nul, int use_float, orc_kernel_result* out) {
  std::memset(out, 0, sizeof(*out));
  F32 F{use_float != 0};
  // genotype = getFlippedToMinorPolymorphicGenotype
  std::vector<double> Gf((size_t)N * M);
  std::vector<int> fl(M), kp(M);
  const int m = orc_flip_poly(Gp, N, M, Gf.data(), fl.data(), kp.data());
  out->n_poly = m;
  if (m == 0) return -1;
  Mat G = wrapd(Gf.data(), N, m);
  Mat U = wrapd(Up, N, N), X = wrapd(Xp, N, d);
  const double sigma2 = nul->sigma2, delta = nul->delta;
  // Sigma, SigmaInv (raw S)
  Mat Sig(N, N), Sinv(N, N);
  for (int64_t a = 0; a < N; ++a)
    for (int64_t b = 0; b <= a; ++b) {
      double s1 = 0, s2 = 0;
      for (int64_t k = 0; k < N; ++k) {
        const double uu = F(F(U(a, k)) * F(U(b, k)));
        s1 = F(s1 + F(uu * F(F(S[k]) + delta)));
        s2 = F(s2 + F(uu * F(1.0 / F(F(S[k]) + delta))));
      }
      Sig(a, b) = Sig(b, a) = F(s1 * sigma2);
      Sinv(a, b) = Sinv(b, a) = F(s2 / sigma2);
    }
  // C = X' SigmaInv X, Cinv
  Mat SX(N, d);
  for (int64_t i = 0; i < N; ++i)
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int64_t k = 0; k < N; ++k) s = F(s + F(Sinv(i, k) * F(X(k, a))));
      SX(i, a) = s;
    }
  Mat C(d, d), I(d, d), Cinv;
  for (int a = 0; a < d; ++a) {
    I(a, a) = 1.0;
    for (int b = 0; b < d; ++b) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(F(X(i, a)) * SX(i, b)));
      C(a, b) = s;
    }
  }
  if (!orc::sym_solve(C, I, &Cinv)) return -1;
  // Sinv_resid = SigmaInv (y - X beta)
  std::vector<double> r(N), sr(N);
  for (int64_t i = 0; i < N; ++i) {
    double p = 0;
    for (int a = 0; a < d; ++a) p = F(p + F(F(X(i, a)) * F(nul->beta[a])));
    r[i] = F(F(y[i]) - p);
  }
  for (int64_t i = 0; i < N; ++i) {
    double s = 0;
    for (int64_t k = 0; k < N; ++k) s = F(s + F(Sinv(i, k) * r[k]));
    sr[i] = s;
  }
  // weights: beta_pdf(FastGetAF; 1, 25), FastGetAF = 0.5 * alpha.g / denom with |lambda| only
  std::vector<double> u1(N), alpha(N);
  for (int64_t k = 0; k < N; ++k) {
    double s = 0;
    for (int64_t i = 0; i < N; ++i) s = F(s + F(U(i, k)));
    u1[k] = s;
  }
  double denom = 0;
  for (int64_t k = 0; k < N; ++k) denom = F(denom + F(F(u1[k] / std::fabs(F(S[k]))) * u1[k]));
  for (int64_t i = 0; i < N; ++i) {
    double s = 0;
    for (int64_t k = 0; k < N; ++k) s = F(s + F(F(u1[k] * F(1.0 / std::fabs(F(S[k])))) * F(U(i, k))));
    alpha[i] = s;
  }
  std::vector<double> w(m);
  for (int j = 0; j < m; ++j) {
    double af = 0.0;
    if (denom != 0.0) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(alpha[i] * F(G(i, j))));
      af = 0.5 * (s / denom);
    }
    w[j] = F(orc_beta_pdf(af, 1.0, 25.0));
  }
  // wg = diag(w) G'; Q = ||wg Sinv_resid||^2
  double Q = 0;
  for (int j = 0; j < m; ++j) {
    double s = 0;
    for (int64_t i = 0; i < N; ++i) s = F(s + F(F(w[j] * F(G(i, j))) * sr[i]));
    Q = F(Q + F(s * s));
  }
  // K = wg P0 wg', P0 = Sigma - X Cinv X'
  Mat XC(N, d);  // X Cinv
  for (int64_t i = 0; i < N; ++i)
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int b = 0; b < d; ++b) s = F(s + F(F(X(i, b)) * F(Cinv(b, a))));
      XC(i, a) = s;
    }
  Mat PG(N, m);  // P0 * (w_j g_j)
  for (int j = 0; j < m; ++j) {
    std::vector<double> xtg(d, 0.0);
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(F(X(i, a)) * F(w[j] * F(G(i, j)))));
      xtg[a] = s;
    }
    for (int64_t i = 0; i < N; ++i) {
      double s = 0;
      for (int64_t k = 0; k < N; ++k) s = F(s + F(Sig(i, k) * F(w[j] * F(G(k, j)))));
      double t = 0;
      for (int a = 0; a < d; ++a) t = F(t + F(XC(i, a) * xtg[a]));
      PG(i, j) = F(s - t);
    }
  }
  Mat K(m, m);
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(F(w[a] * F(G(i, a))) * PG(i, b)));
      K(a, b) = s;
    }
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < a; ++b) K(a, b) = K(b, a) = 0.5 * (K(a, b) + K(b, a));
  std::vector<double> ev = orc::sym_eigvals(K);
  const int r_ub = (int)std::min<int64_t>(N, m);
  int nl = 0;
  for (int i = (int)ev.size() - 1; i >= 0; --i) {
    if (F(ev[i]) > 1e-30 && nl < r_ub)
      out->lambda[nl++] = F(ev[i]);
    else
      break;
  }
  out->n_lambda = nl;
  out->Q = Q;
  int fault = 0;
  out->pvalue = orc_davies_pvalue(out->lambda, nl, Q, &fault);
  out->fit_ok = 1;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// MetaCovFamQtl (src/Model.cpp:437-504): g <- U'(g - mean(g)) (FastLMM::TransformCentered, FastLMM.cpp:611-625);
// covXX = sum g1 g2 / (lambda + delta) / sigma2 (:535-549); covXZ = g' diag(1/(lambda+delta)) ux / sigma2 (:568-590);
// covZZ = ux' diag(1/(lambda+delta)) ux / sigma2 (:510-516), lambda = |S|; rows and value as in orc_metacov.
// ---------------------------------------------------------------------------------------------
int orc_metacov_fam(const double* Gp, int64_t N, int V, const int* chrom, const int* pos, const double* Xp, int d,
                    const double* Up, const double* S, const orc_fam_null* nul, int window, int use_float, int* kept,
                    double* cov, int* row_end, double* xz, double* zz) {
  F32 F{use_float != 0};
  Mat G = wrapd(Gp, N, V), U = wrapd(Up, N, N), X = wrapd(Xp, N, d);
  const double sigma2 = nul->sigma2, delta = nul->delta;
  std::vector<double> w(N);  // 1 / (lambda + delta)
  for (int64_t i = 0; i < N; ++i) w[i] = F(1.0 / F(std::fabs(F(S[i])) + delta));
  Mat ux(N, d);
  for (int64_t k = 0; k < N; ++k)
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(F(U(i, k)) * F(X(i, a))));
      ux(k, a) = s;
    }
  Mat ZZ(d, d), I(d, d), ZZinv;
  for (int a = 0; a < d; ++a) {
    I(a, a) = 1.0;
    for (int b = 0; b < d; ++b) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(F(ux(i, a) * w[i]) * ux(i, b)));
      ZZ(a, b) = F(s / sigma2);
    }
  }
  if (!orc::sym_solve(ZZ, I, &ZZinv)) return -1;
  for (int a = 0; a < d; ++a)
    for (int b = 0; b < d; ++b) zz[a * d + b] = ZZ(a, b);
  std::vector<std::vector<double>> gt(V);
  for (int j = 0; j < V; ++j) {
    bool mono = true;
    for (int64_t i = 1; i < N; ++i)
      if (G(i, j) != G(0, j)) {
        mono = false;
        break;
      }
    kept[j] = mono ? 0 : 1;
    row_end[j] = -1;
    for (int k = 0; k < d; ++k) xz[(size_t)j * d + k] = NAN;
    if (mono) continue;
    std::vector<double> g(N);
    double sx = 0;
    for (int64_t i = 0; i < N; ++i) {
      g[i] = F(G(i, j));
      sx = F(sx + g[i]);
    }
    const double avg = F(sx / (double)N);
    for (int64_t i = 0; i < N; ++i) g[i] = F(g[i] - avg);
    gt[j].assign(N, 0.0);
    for (int64_t k = 0; k < N; ++k) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(F(U(i, k)) * g[i]));
      gt[j][k] = s;
    }
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(F(gt[j][i] * w[i]) * ux(i, a)));
      xz[(size_t)j * d + a] = F(s / sigma2);
    }
  }
  for (size_t i = 0; i < (size_t)V * V; ++i) cov[i] = NAN;
  for (int h = 0; h < V; ++h) {
    if (!kept[h]) continue;
    for (int j = h; j < V; ++j) {
      if (chrom[j] != chrom[h] || std::abs(pos[j] - pos[h]) > window) break;
      if (!kept[j]) continue;
      double xx = 0;
      for (int64_t i = 0; i < N; ++i) xx = F(xx + F(F(gt[h][i] * w[i]) * gt[j][i]));
      xx = F(xx / sigma2);
      double quad = 0;
      for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b)
          quad = F(quad + F(F(xz[(size_t)h * d + a] * F(ZZinv(a, b))) * xz[(size_t)j * d + b]));
      cov[h + (size_t)j * V] = F(xx - quad);
      row_end[h] = j;
    }
  }
  return 0;
}

int orc_fam_burden(const double* Gp, int64_t N, int M, const double* Xp, const double* y, int d, const double* Up,
                   const double* S, const orc_fam_null* nul, int which, int use_float, orc_fam_burden_result* out) {
  std::memset(out, 0, sizeof(*out));
  F32 F{use_float != 0};
  std::vector<double> Gf((size_t)N * M);
  std::vector<int> fl(M), kp(M);
  std::vector<double> c(N, 0.0);
  if (which >= 2) {
    // MetaScoreTest with kinship (MetaFamQtl::TestCovariate, src/Model.h:3421-3434): the single genotype column as it
    // is (imputed, not flipped); monomorphic sites are skipped before the test (:3246-3250)
    if (M != 1) return -1;
    bool mono = true;
    for (int64_t i = 1; i < N; ++i)
      if (Gp[i] != Gp[0]) mono = false;
    out->num_site = mono ? 0 : 1;
    if (mono) return -1;
    for (int64_t i = 0; i < N; ++i) c[i] = Gp[i];
  } else {
    const int m = orc_flip_poly(Gp, N, M, Gf.data(), fl.data(), kp.data());
    out->num_site = m;
    if (m == 0) return -1;
    // cmcCollapse / zegginiCollapse (src/Model.cpp:73-89,115-130)
    for (int64_t i = 0; i < N; ++i) {
      int n = 0;
      for (int j = 0; j < m; ++j)
        if ((int)Gf[(size_t)j * N + i] > 0) ++n;
      c[i] = which == 0 ? (n > 0 ? 1.0 : 0.0) : (double)n;
    }
  }
  Mat U = wrapd(Up, N, N), X = wrapd(Xp, N, d);
  const double sigma2 = nul->sigma2, delta = nul->delta;
  std::vector<double> lam(N), sinv(N);
  for (int64_t i = 0; i < N; ++i) {
    lam[i] = std::fabs(F(S[i]));
    sinv[i] = F(1.0 / F(lam[i] + delta));
  }
  // ux = U'X, uResid = U'y - ux beta, u_g_center = U'(g - mean(g))
  Mat ux(N, d);
  std::vector<double> ur(N), ug(N), ugc(N);
  double cs = 0;
  for (int64_t i = 0; i < N; ++i) cs = F(cs + F(c[i]));
  // which == 3: MetaFamBinary calls lmm.disableCenterGenotype() (src/Model.h:3558-3560; FastLMM.cpp:218-223)
  const double cmean = (which == 3) ? 0.0 : F(cs / (double)N);
  for (int64_t k = 0; k < N; ++k) {
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(F(U(i, k)) * F(X(i, a))));
      ux(k, a) = s;
    }
    double sy = 0, sg = 0, sc = 0;
    for (int64_t i = 0; i < N; ++i) {
      sy = F(sy + F(F(U(i, k)) * F(y[i])));
      sg = F(sg + F(F(U(i, k)) * F(c[i])));
      sc = F(sc + F(F(U(i, k)) * F(F(c[i]) - cmean)));
    }
    double p = 0;
    for (int a = 0; a < d; ++a) p = F(p + F(ux(k, a) * F(nul->beta[a])));
    ur[k] = F(sy - p);
    ug[k] = sg;
    ugc[k] = sc;
  }
  // scaledK = Sinv - Sinv ux (ux' Sinv ux)^-1 ux' Sinv   (N x N, FastLMM.cpp:124-138)
  Mat A(d, d), I(d, d), Ai;
  for (int a = 0; a < d; ++a) {
    I(a, a) = 1.0;
    for (int b = 0; b < d; ++b) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(F(ux(i, a) * sinv[i]) * ux(i, b)));
      A(a, b) = s;
    }
  }
  if (!orc::sym_solve(A, I, &Ai)) return -1;
  Mat K(N, N);
  for (int64_t i = 0; i < N; ++i)
    for (int64_t j = 0; j < N; ++j) {
      double q = 0;
      for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) q = F(q + F(F(F(sinv[i] * ux(i, a)) * F(Ai(a, b))) * F(ux(j, b) * sinv[j])));
      K(i, j) = F((i == j ? sinv[i] : 0.0) - q);
    }
  double Us = 0;
  for (int64_t i = 0; i < N; ++i) Us = F(Us + F(F(ugc[i] * ur[i]) / F(lam[i] + delta)));
  Us = Us / sigma2;
  double Vs = 0;
  for (int64_t i = 0; i < N; ++i) {
    double t = 0;
    for (int64_t j = 0; j < N; ++j) t = F(t + F(K(i, j) * ugc[j]));
    Vs = F(Vs + F(ugc[i] * t));
  }
  Vs = Vs / sigma2;
  out->U = Us;
  out->V = Vs;
  if (Vs > 0.0) {
    out->stat = Us * Us / Vs;
    out->pvalue = orc_chisq_Q(out->stat, 1.0);
  } else {
    out->stat = 0;
    out->pvalue = 1.0;
  }
  // GetAF: 0.5 * (u1s . ug) / (u1s . u1), u1 = U'1, u1s = u1 / |lambda|
  double denom = 0, numer = 0;
  for (int64_t k = 0; k < N; ++k) {
    double u1 = 0;
    for (int64_t i = 0; i < N; ++i) u1 = F(u1 + F(U(i, k)));
    const double u1s = F(u1 / lam[k]);
    denom = F(denom + F(u1s * u1));
    numer = F(numer + F(u1s * ug[k]));
  }
  out->af = (denom == 0.0) ? 0.0 : 0.5 * (numer / denom);
  out->fit_ok = 1;
  return 0;
}

// FastLMM::GetNullCovB (regression/FastLMM.cpp:473-483): (ux' diag(lambda + delta) ux)^-1, ux = U'X, lambda = |S|
// (the reference multiplies by lambda + delta although its comment derives the inverse weights; restated literally).
int orc_fastlmm_covb(const double* Xp, int64_t N, int d, const double* Up, const double* S, double delta, int use_float,
                     double* covb) {
  F32 F{use_float != 0};
  Mat U = wrapd(Up, N, N), X = wrapd(Xp, N, d), ux(N, d);
  for (int64_t k = 0; k < N; ++k)
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(F(U(i, k)) * F(X(i, a))));
      ux(k, a) = s;
    }
  Mat A(d, d), I(d, d), Ai;
  for (int a = 0; a < d; ++a) {
    I(a, a) = 1.0;
    for (int b = 0; b < d; ++b) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(F(ux(i, a) * F(std::fabs(F(S[i])) + delta)) * ux(i, b)));
      A(a, b) = s;
    }
  }
  if (!orc::sym_solve(A, I, &Ai)) return -1;
  for (int a = 0; a < d * d; ++a) covb[a] = Ai.a[a];
  return 0;
}

static double obtain_b_integrand(double t, void* p) {
  const double alpha = *(const double*)p;
  if (t <= 0.0) return 0.0;
  const double x = (1.0 - t) / t;
  auto f = [&](double xx) {  // fIntegrand, src/Model.cpp:339-348
    if (xx > 500 || xx < -500) return 0.0;
    const double tmp = std::exp(alpha + xx);
    if (!std::isfinite(tmp)) return 0.0;  // (alpha = 500, the no-control case, overflows in the reference: NaN there)
    const double k = 1.0 / std::sqrt(2.0 * 3.1415926535897);
    return tmp / (1. + tmp) / (1. + tmp) * k * std::exp(-xx * xx * 0.5);
  };
  return (f(x) + f(-x)) / (t * t);
}

double orc_obtain_b(double alpha) {
  double res = 0, err = 0;
  int neval = 0;
  orc_qags(obtain_b_integrand, &alpha, 0.0, 1.0, 0.0, 1e-10, 1000, &res, &err, &neval);
  return res;
}

int orc_metacov_fam_binary(const double* Gp, int64_t N, int V, const int* chrom, const int* pos, const double* Xp,
                           int d, const double* y, const double* Up, const double* S, const orc_fam_null* nul,
                           int window, int use_float, int* kept, double* cov, int* row_end, double* xz, double* zz) {
  int nCase = 0, nCtrl = 0;
  for (int64_t i = 0; i < N; ++i) {
    if (y[i] == 1) ++nCase;
    else if (y[i] == 0) ++nCtrl;
  }
  const float alpha = (nCtrl > 0) ? (float)std::log(1.0 * nCase / nCtrl) : 500.f;  // `float alpha` member
  const float b = (float)orc_obtain_b((double)alpha);                              // `float b` member
  const int rc = orc_metacov_fam(Gp, N, V, chrom, pos, Xp, d, Up, S, nul, window, use_float, kept, cov, row_end, xz, zz);
  if (rc) return rc;
  // covXX, covXZ, covZZ are each multiplied by b*b (Model.cpp:651-668); value = xx - xz' zz^-1 xz scales by b^2 too
  const double b2 = (double)b * (double)b;
  for (size_t i = 0; i < (size_t)V * V; ++i) cov[i] *= b2;
  for (size_t i = 0; i < (size_t)V * d; ++i) xz[i] *= b2;
  for (int i = 0; i < d * d; ++i) zz[i] *= b2;
  return 0;
}

}  // extern "C"
