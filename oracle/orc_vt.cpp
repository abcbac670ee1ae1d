// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_api.h).  CPU restatement of AnalyticVT for unrelated samples:
//   AnalyticVT::fit                 src/Model.h:2131-2226   (LITERAL: the N x N residual-forming matrix I - H,
//                                   x = (I - H) G centred, y = (I - H) y, sigma2 = var(y), u = x'y, v = x'x sigma2)
//   calculateResidualMatrix / HatMatrix   regression/LinearRegression.cpp:86-123
//   centerMatrix / getVariance      src/LinearAlgebra.h:51-65,151-165
//   MultivariateVT::compute         regression/MultivariateVT.cpp:22-144
//   getBandProbFromCov / toCor      regression/MultivariateNormalDistribution.cpp:52-69, n == 1 shortcut :18-21
// The band probability itself: the reference calls Genz's MVTDST (regression/libMvtnorm/mvt.f) — a randomised rule
// with abseps 1e-3.  This restatement evaluates the same integral with an INDEPENDENT deterministic method (Genz's
// transformation on a Halton point set with a Cranley-Patterson shift average), and tests/test_vt_cpu.py checks both
// it and the device's lattice rule against the reference's own MVTDST compiled where it lies (oracle/_ref/libref_mvt.so)
// to within that rule's accuracy.  Parity of the p-value beyond ~1e-3 does not exist in the reference.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <set>
#include <vector>
#include "orc_linalg.h"

using orc::Mat;
// DataConsolidator::getFlippedToMinorPolymorphicGenotype restated in orc_models.cpp
extern "C" int orc_flip_poly(const double* Gp, int64_t N, int M, double* out, int* flipped, int* kept);

namespace {
Mat wrap(const double* p, int64_t r, int64_t c) {
  Mat m(r, c);
  std::memcpy(m.a.data(), p, sizeof(double) * (size_t)r * c);
  return m;
}
double phi(double x) { return 0.5 * std::erfc(-x / std::sqrt(2.0)); }
double phiinv(double p) {  // bisection + Newton polish: slow and simple
  if (p <= 0) return -INFINITY;
  if (p >= 1) return INFINITY;
  double lo = -40, hi = 40;
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (phi(mid) < p)
      lo = mid;
    else
      hi = mid;
  }
  return 0.5 * (lo + hi);
}
double halton(long long idx, int base) {
  double f = 1.0, r = 0.0;
  while (idx > 0) {
    f /= base;
    r += f * (double)(idx % base);
    idx /= base;
  }
  return r;
}
}  // namespace

extern "C" {

struct orc_vt_result {
  int fit_ok, n_poly, opt_num, n_cutoff;
  double min_maf, max_maf, opt_maf, U, V, stat, pvalue, p_err;
};

// P(-T < Z_i < T) for correlation R (row-major n x n): Genz transformation, Halton points, 12 random shifts
double orc_mvn_band(const double* R, int n, double T, long long points, double* err_out) {
  if (n == 1) {
    if (err_out) *err_out = 0;
    return phi(T) - phi(-T);
  }
  std::vector<double> L((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j) {
    double s = R[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) s -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    const double l = s > 1e-10 ? std::sqrt(s) : 0.0;
    L[(size_t)j * n + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double t = 0.0;
      if (l > 0) {
        t = R[(size_t)i * n + j];
        for (int k = 0; k < j; ++k) t -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
        t /= l;
      }
      L[(size_t)i * n + j] = t;
    }
  }
  std::vector<int> primes;
  for (int c = 2; (int)primes.size() < n; ++c) {
    bool p = true;
    for (int q = 2; q * q <= c; ++q)
      if (c % q == 0) p = false;
    if (p) primes.push_back(c);
  }
  const int shifts = 12;
  std::vector<double> means(shifts, 0.0), y(n);
  uint64_t state = 88172645463325252ull;
  auto rnd = [&]() {
    state ^= state << 13;
    state ^= state >> 7;
    state ^= state << 17;
    return (double)(state >> 11) / 9007199254740992.0;
  };
  for (int sft = 0; sft < shifts; ++sft) {
    std::vector<double> sh(n);
    for (int i = 0; i < n; ++i) sh[i] = rnd();
    double acc = 0;
    for (long long k = 1; k <= points; ++k) {
      double f = 1.0;
      for (int i = 0; i < n && f > 0; ++i) {
        double s = 0;
        for (int q = 0; q < i; ++q) s += L[(size_t)i * n + q] * y[q];
        const double l = L[(size_t)i * n + i];
        if (l > 0) {
          const double dlo = phi((-T - s) / l), dhi = phi((T - s) / l);
          f *= dhi - dlo;
          if (i + 1 < n) {
            double x = halton(k, primes[i]) + sh[i];
            x -= std::floor(x);
            x = std::fabs(2 * x - 1);
            x = std::fmin(std::fmax(x, 1e-15), 1 - 1e-15);
            y[i] = phiinv(dlo + x * (dhi - dlo));
          }
        } else {
          if (!(s > -T && s < T)) f = 0;
          y[i] = 0;
        }
      }
      acc += f;
    }
    means[sft] = acc / (double)points;
  }
  double m = 0, sq = 0;
  for (double v : means) {
    m += v;
    sq += v * v;
  }
  m /= shifts;
  const double var = std::fmax(0.0, sq / shifts - m * m) / (shifts - 1);
  if (err_out) *err_out = 3.5 * std::sqrt(var);
  return m;
}

static int vt_compute(int m, const double* af, const Mat& U, const Mat& V, long long mvn_points, orc_vt_result* out,
                      double* cor_out);

// G: imputed, unflipped N x M (column-major); af: M counter frequencies (dc->getMarkerFrequency); X: N x d with the
// intercept; y: N.  cor_out (optional, K x K row-major, K <= M) receives the correlation of the threshold statistics.
int orc_analytic_vt(const double* Gp, const double* af, const double* Xp, const double* yp, int64_t N, int M, int d,
                    long long mvn_points, orc_vt_result* out, double* cor_out) {
  std::memset(out, 0, sizeof(*out));
  std::vector<double> gbuf((size_t)N * M);
  const int m = orc_flip_poly(Gp, N, M, gbuf.data(), nullptr, nullptr);
  Mat G = wrap(gbuf.data(), N, m);
  out->n_poly = m;
  if (m == 0) return -1;
  Mat X = wrap(Xp, N, d);
  // residual-forming matrix (I - H), H = X (X'X)^-1 X'
  Mat XtX = orc::AtB(X, X), I(d, d), XtXinv;
  for (int k = 0; k < d; ++k) I(k, k) = 1.0;
  if (!orc::chol_solve(XtX, I, &XtXinv)) return -1;
  Mat H = orc::mul(orc::mul(X, XtXinv), orc::transpose(X));
  Mat Rm(N, N);
  for (int64_t i = 0; i < N; ++i)
    for (int64_t j = 0; j < N; ++j) Rm(i, j) = (i == j) ? 1.0 - H(i, j) : -H(i, j);
  Mat yv = wrap(yp, N, 1);
  Mat y = orc::mul(Rm, yv);
  Mat x = orc::mul(Rm, G);
  for (int j = 0; j < m; ++j) {  // centerMatrix
    double s = 0;
    for (int64_t i = 0; i < N; ++i) s += x(i, j);
    s /= (double)N;
    for (int64_t i = 0; i < N; ++i) x(i, j) -= s;
  }
  double sigma2;
  {
    double s = 0;
    for (int64_t i = 0; i < N; ++i) s += y(i, 0);
    const double avg = s / (double)N;
    s = 0;
    for (int64_t i = 0; i < N; ++i) s += (y(i, 0) - avg) * (y(i, 0) - avg);
    sigma2 = s / (double)N;
  }
  Mat U = orc::AtB(x, y), V = orc::AtB(x, x);
  for (double& e : V.a) e *= sigma2;
  return vt_compute(m, af, U, V, mvn_points, out, cor_out);
}

// MultivariateVT::compute (regression/MultivariateVT.cpp:22-144) + the p-value; af: numFreq frequencies, U: numFreq x 1,
// V: numFreq x numFreq
static int vt_compute(int m, const double* af, const Mat& U, const Mat& V, long long mvn_points, orc_vt_result* out,
                      double* cor_out) {
  // ---- MultivariateVT::compute ----
  const int numFreq = m;
  std::vector<double> maf(numFreq);
  std::set<int> skip, freqTable;
  int numKeep = 0;
  for (int i = 0; i < numFreq; ++i) {
    maf[i] = af[i] < 0.5 ? af[i] : 1.0 - af[i];  // (filtered position i reads the counter of unfiltered column i)
    if (maf[i] < 1e-10) {
      skip.insert(i);
      continue;
    }
    if (V(i, i) < 1e-10) {
      skip.insert(i);
      continue;
    }
    const int mafInt = (int)std::ceil(maf[i] * 1000000);
    if (freqTable.count(mafInt)) continue;
    freqTable.insert(mafInt);
    numKeep++;
  }
  if (numKeep == 0) return -1;
  std::vector<double> cutoff;
  for (int k : freqTable) cutoff.push_back(1.0 * k / 1000000);
  Mat phi_(numFreq, numKeep);
  for (int i = 0; i < numFreq; ++i)
    for (int j = 0; j < numKeep; ++j) phi_(i, j) = (!skip.count(i) && maf[i] <= cutoff[j]) ? 1.0 : 0.0;
  Mat u_phi = orc::AtB(U, phi_);                      // 1 x K
  Mat v_phi = orc::AtB(phi_, orc::mul(V, phi_));      // K x K
  int maxIdx = -1;
  double maxVal = -DBL_MAX;
  for (int i = 0; i < numKeep; ++i) {
    const double t = std::fabs(u_phi(0, i) / std::sqrt(v_phi(i, i)));
    if (t > maxVal) {
      maxIdx = i;
      maxVal = t;
    }
  }
  if (maxIdx < 0) return -1;
  out->min_maf = maf[0];
  out->max_maf = maf[0];
  for (double f : maf) {
    out->min_maf = std::fmin(out->min_maf, f);
    out->max_maf = std::fmax(out->max_maf, f);
  }
  out->opt_maf = cutoff[maxIdx];
  out->U = u_phi(0, maxIdx);
  out->V = v_phi(maxIdx, maxIdx);
  out->stat = maxVal;
  out->n_cutoff = numKeep;
  for (int i = 0; i < numFreq; ++i)
    if (phi_(i, maxIdx) > 0) ++out->opt_num;
  std::vector<double> cor((size_t)numKeep * numKeep);
  for (int i = 0; i < numKeep; ++i)
    for (int j = 0; j < numKeep; ++j)
      cor[(size_t)i * numKeep + j] = (i == j) ? 1.0 : v_phi(i, j) / std::sqrt(v_phi(i, i) * v_phi(j, j));
  if (cor_out) std::memcpy(cor_out, cor.data(), sizeof(double) * cor.size());
  double err = 0;
  const double prob = orc_mvn_band(cor.data(), numKeep, maxVal, mvn_points, &err);
  out->pvalue = 1.0 - prob;
  out->p_err = err;
  out->fit_ok = 1;
  return 0;
}

// FamAnalyticVT (src/Model.h:2189-2214): af_i = FastLMM::FastGetAF (regression/FastLMM.cpp:402-443) of column i of the
// flipped, polymorphic genotype; (u, v) = FastLMM::CalculateUandV (:259-291) with the LITERAL N x N scaledK (:124-138):
//   u = (U'g_c)' (lambda + delta)^-1 uResid / sigma2,  v = (U'g_c)' scaledK (U'g_c) / sigma2,  lambda = |S|.
// delta, sigma2, beta: the FastLMM null model (orc_fastlmm_null).
int orc_fam_analytic_vt(const double* Gp, const double* Xp, const double* yp, int64_t N, int M, int d, const double* Up,
                        const double* S, double delta, double sigma2, const double* beta, long long mvn_points,
                        orc_vt_result* out, double* cor_out) {
  std::memset(out, 0, sizeof(*out));
  std::vector<double> gbuf((size_t)N * M);
  const int m = orc_flip_poly(Gp, N, M, gbuf.data(), nullptr, nullptr);
  out->n_poly = m;
  if (m == 0) return -1;
  Mat G = wrap(gbuf.data(), N, m), X = wrap(Xp, N, d), Um = wrap(Up, N, N);
  std::vector<double> lam(N), sinv(N);
  for (int64_t i = 0; i < N; ++i) {
    lam[i] = std::fabs(S[i]);
    sinv[i] = 1.0 / (lam[i] + delta);
  }
  Mat ux = orc::AtB(Um, X);                         // U'X
  Mat uy = orc::AtB(Um, wrap(yp, N, 1));
  std::vector<double> ur(N);
  for (int64_t k = 0; k < N; ++k) {
    double p = 0;
    for (int a = 0; a < d; ++a) p += ux(k, a) * beta[a];
    ur[k] = uy(k, 0) - p;
  }
  Mat A(d, d), I(d, d), Ai;                         // (ux' Sinv ux)^-1
  for (int a = 0; a < d; ++a) {
    I(a, a) = 1.0;
    for (int b = 0; b < d; ++b) {
      double sacc = 0;
      for (int64_t i = 0; i < N; ++i) sacc += ux(i, a) * sinv[i] * ux(i, b);
      A(a, b) = sacc;
    }
  }
  if (!orc::sym_solve(A, I, &Ai)) return -1;
  Mat K(N, N);                                      // scaledK = Sinv - Sinv ux (ux' Sinv ux)^-1 ux' Sinv
  for (int64_t i = 0; i < N; ++i)
    for (int64_t j = 0; j < N; ++j) {
      double q = 0;
      for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) q += sinv[i] * ux(i, a) * Ai(a, b) * ux(j, b) * sinv[j];
      K(i, j) = (i == j ? sinv[i] : 0.0) - q;
    }
  Mat Gc = G;                                       // g.rowwise() - g.colwise().mean()
  for (int j = 0; j < m; ++j) {
    double sacc = 0;
    for (int64_t i = 0; i < N; ++i) sacc += G(i, j);
    sacc /= (double)N;
    for (int64_t i = 0; i < N; ++i) Gc(i, j) -= sacc;
  }
  Mat ugc = orc::AtB(Um, Gc), ug = orc::AtB(Um, G); // U'g_c, U'g
  Mat Uv(m, 1), V(m, m);
  for (int j = 0; j < m; ++j) {
    double sacc = 0;
    for (int64_t i = 0; i < N; ++i) sacc += ugc(i, j) * sinv[i] * ur[i];
    Uv(j, 0) = sacc / sigma2;
  }
  Mat Kg = orc::mul(K, ugc);
  Mat Vm = orc::AtB(ugc, Kg);
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) V(a, b) = Vm(a, b) / sigma2;
  // FastGetAF: 0.5 * (u1s . U'g_j) / (u1s . u1), u1 = U'1, u1s = u1 / lambda
  std::vector<double> af(m);
  {
    std::vector<double> u1(N), u1s(N);
    double denom = 0;
    for (int64_t k = 0; k < N; ++k) {
      double sacc = 0;
      for (int64_t i = 0; i < N; ++i) sacc += Um(i, k);
      u1[k] = sacc;
      u1s[k] = sacc / lam[k];
      denom += u1s[k] * u1[k];
    }
    for (int j = 0; j < m; ++j) {
      double numer = 0;
      for (int64_t k = 0; k < N; ++k) numer += u1s[k] * ug(k, j);
      af[j] = denom == 0.0 ? 0.0 : 0.5 * (numer / denom);
    }
  }
  return vt_compute(m, af.data(), Uv, V, mvn_points, out, cor_out);
}

}  // extern "C"
