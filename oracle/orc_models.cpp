// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// CPU restatement of the per-gene models on the rvtests kernel/burden hot path.  Each function
// names the reference lines it follows.  Parity status: the reference holds NO unit test or golden
// output for Skat / SkatO / CMC / Zeggini, and Eigen 3.3.9 is not vendored so these
// bodies cannot be compiled from /root/reference; their scalar building blocks (Davies, Liu, GSL
// special functions, QAGS) ARE pinned against compiled reference code, the model bodies are pinned
// against an independent numpy/scipy restatement (tests/golden/make_model_golden.py).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>
#include "orc_api.h"
#include "orc_linalg.h"

using orc::Mat;

namespace {

Mat wrap(const double* p, int64_t r, int64_t c) {
  Mat m(r, c);
  std::memcpy(m.a.data(), p, sizeof(double) * (size_t)r * c);
  return m;
}

// ---------------------------------------------------------------------------------------------
// DataConsolidator::getFlippedToMinorPolymorphicGenotype  (src/DataConsolidator.h:128-132):
//   convertToMinorAlleleCount (src/DataConsolidator.cpp:46-69) then removeMonomorphicMarker (:94-142)
// ---------------------------------------------------------------------------------------------
Mat flip_poly(const Mat& G, std::vector<int>* flipped, std::vector<int>* kept) {
  const int64_t N = G.r;
  const int M = (int)G.c;
  Mat F(N, M);
  flipped->assign(M, 0);
  kept->assign(M, 1);
  for (int j = 0; j < M; ++j) {
    double s = 0;
    for (int64_t i = 0; i < N; ++i) s += G(i, j);
    if (s <= (double)N) {
      for (int64_t i = 0; i < N; ++i) F(i, j) = G(i, j);
    } else {
      (*flipped)[j] = 1;
      for (int64_t i = 0; i < N; ++i) F(i, j) = 2 - G(i, j);
    }
  }
  // isMonomorphicMarker on the flipped matrix
  int mout = 0;
  for (int j = 0; j < M; ++j) {
    int64_t first = N;
    for (int64_t i = 0; i < N; ++i)
      if (F(i, j) >= 0) {
        first = i;
        break;
      }
    bool mono = true;
    for (int64_t r = first + 1; r < N; ++r) {
      if (F(r, j) < 0) continue;
      if (F(r, j) != F(first, j)) {
        mono = false;
        break;
      }
    }
    (*kept)[j] = mono ? 0 : 1;
    if (!mono) ++mout;
  }
  Mat O(N, mout);
  int c = 0;
  for (int j = 0; j < M; ++j)
    if ((*kept)[j]) {
      std::memcpy(O.col(c), F.col(j), sizeof(double) * (size_t)N);
      ++c;
    }
  return O;
}

// weights: src/Model.h:2644-2661 (SKAT: beta_pdf^2) and :2799-2813 (SKAT-O: beta_pdf).
// Quirk #3: weight i reads counter[i] of the UNFILTERED column order.
std::vector<double> beta_weights(const double* af, int mpoly, double b1, double b2, bool squared) {
  std::vector<double> w(mpoly);
  for (int i = 0; i < mpoly; ++i) {
    double freq = af[i];
    if (freq > 0.5) freq = 1.0 - freq;
    if (freq > 1e-30) {
      w[i] = orc_beta_pdf(freq, b1, b2);
      if (squared) w[i] *= w[i];
    } else
      w[i] = 0.0;
  }
  return w;
}

// Skat.cpp:87-103 — eigenvalue filter + Davies with Liu fallback
void skat_pvalue(const std::vector<double>& evals_asc, int64_t N, int M, double Q, orc_kernel_result* out) {
  const int r_ub = (int)std::min<int64_t>(N, M);
  int r = 0;
  std::vector<double> lam;
  for (int i = (int)evals_asc.size() - 1; i >= 0; --i) {
    if (evals_asc[i] > 1e-30 && r < r_ub) {
      lam.push_back(evals_asc[i]);
      ++r;
    } else
      break;
  }
  out->n_lambda = r;
  for (int i = 0; i < r && i < 512; ++i) out->lambda[i] = lam[i];
  double p = orc_davies_pvalue(lam.data(), r, Q, nullptr);
  if (p <= 0.0 || p == 1.0) p = orc_liu_pvalue(lam.data(), r, Q);
  out->pvalue = p;
}

// ---- SKAT-O helpers -------------------------------------------------------------------------
struct Moment {
  double muQ, varQ, df;
};

// SkatO.cpp:350-382
int get_eigen(const Mat& K, std::vector<double>* lambda) {
  std::vector<double> values = orc::sym_eigvals(K);
  const int n = (int)values.size();
  int numNonZero = 0;
  double sumNonZero = 0.;
  for (int i = 0; i < n; ++i)
    if (values[i] > 0) {
      ++numNonZero;
      sumNonZero += values[i];
    }
  if (numNonZero == 0) return -1;
  const double t = sumNonZero / numNonZero / 100000;
  int numKeep = n;
  for (int i = 0; i < n; ++i) {
    if (values[i] < t)
      --numKeep;
    else
      break;
  }
  lambda->resize(numKeep);
  for (int i = 0; i < numKeep; ++i) (*lambda)[i] = values[n - 1 - i];
  return 0;
}

// SkatO.cpp:383-418
void get_moment(const std::vector<double>& la, Moment* m) {
  double c[4] = {0, 0, 0, 0};
  for (double l : la) {
    c[0] += l;
    c[1] += l * l;
    c[2] += (l * l) * l;
    c[3] += (l * l) * (l * l);
  }
  m->muQ = c[0];
  const double sigmaQ = std::sqrt(2 * c[1]);
  const double s1 = c[2] / c[1] / std::sqrt(c[1]);
  const double s2 = c[3] / (c[1] * c[1]);
  double a, d, l;
  if (s1 * s1 > s2) {
    a = 1 / (s1 - std::sqrt(s1 * s1 - s2));
    d = (s1 * a - 1.0 * a * a);
    l = a * a - 2 * d;
  } else {
    l = 1. / s2;
    a = std::sqrt(l);
    d = 0;
  }
  (void)a;
  m->varQ = sigmaQ * sigmaQ;
  m->df = l;
}

struct SkatOState {
  int nRho;
  double rhos[11], Qs_minP[11], taus[11];
  double MuQ, VarQ, VarZeta, Df;
  std::vector<double> lambda;
  double lambda_sum;
};

double pval_davies(double Q, const std::vector<double>& lam) {
  return orc_davies_pvalue(lam.data(), (int)lam.size(), Q, nullptr);
}

// SkatO.cpp:303-325
double integrand_davies(double x, void* param) {
  SkatOState* s = (SkatOState*)param;
  double kappa = DBL_MAX;
  for (int i = 0; i < s->nRho; ++i) {
    const double v = (s->Qs_minP[i] - s->taus[i] * x) / (1.0 - s->rhos[i]);
    if (i == 0) kappa = v;
    if (v < kappa) kappa = v;
  }
  double temp;
  if (kappa > s->lambda_sum * 10000) {
    temp = 0.0;
  } else {
    const double Q = (kappa - s->MuQ) * std::sqrt(s->VarQ - s->VarZeta) / std::sqrt(s->VarQ) + s->MuQ;
    temp = pval_davies(Q, s->lambda);
    if (temp <= 0.0 || temp == 1.0) temp = orc_liu_pvalue(s->lambda.data(), (int)s->lambda.size(), Q);
  }
  return (1.0 - temp) * orc_chisq_pdf(x, 1.0);
}

// SkatO.cpp:327-337
double integrand_liu(double x, void* param) {
  SkatOState* s = (SkatOState*)param;
  double kappa = DBL_MAX;
  for (int i = 0; i < s->nRho; ++i) {
    const double v = (s->Qs_minP[i] - s->taus[i] * x) / (1.0 - s->rhos[i]);
    if (v < kappa) kappa = v;
  }
  const double Q = (kappa - s->MuQ) / std::sqrt(s->VarQ) * std::sqrt(2.0 * s->Df) + s->Df;
  return orc_chisq_P(Q, s->Df) * orc_chisq_pdf(x, 1.0);
}

// glibc rand() (TYPE_3 additive feedback, r[i] = r[i-3] + r[i-31]); the reference never calls
// srand (src/LinearAlgebra.h:8-21), i.e. the stream of seed 1.
struct GlibcRand {
  uint32_t ring[31];
  int pos;
  void seed(unsigned s) {
    int32_t st[34];
    st[0] = (int32_t)(s == 0 ? 1 : s);
    for (int i = 1; i < 31; i++) {
      int64_t v = (16807LL * st[i - 1]) % 2147483647;
      if (v < 0) v += 2147483647;
      st[i] = (int32_t)v;
    }
    std::vector<uint32_t> o(344);
    for (int i = 0; i < 31; i++) o[i] = (uint32_t)st[i];
    for (int i = 31; i < 34; i++) o[i] = o[i - 31];
    for (int i = 34; i < 344; i++) o[i] = o[i - 31] + o[i - 3];
    for (int i = 0; i < 31; i++) ring[i] = o[344 - 31 + i];
    pos = 0;
  }
  int next() {
    // ring[pos] holds o[k-31]; o[k-3] is 28 slots ahead
    const uint32_t v = ring[pos] + ring[(pos + 28) % 31];
    ring[pos] = v;
    pos = (pos + 1) % 31;
    return (int)(v >> 1);
  }
};
GlibcRand g_rand = [] {
  GlibcRand g;
  g.seed(1);
  return g;
}();

}  // namespace

extern "C" {

void orc_rand_seed(unsigned seed) { g_rand.seed(seed); }
int orc_rand(void) { return g_rand.next(); }

// ---------------------------------------------------------------------------------------------
// LinearRegression::FitLinearModel  (regression/LinearRegression.cpp:20-69)
// ---------------------------------------------------------------------------------------------
int orc_fit_linear(const double* Xp, const double* y, int64_t N, int d, double* beta, double* pred,
                   double* resid, double* sigma2) {
  Mat X = wrap(Xp, N, d);
  Mat Y = wrap(y, N, 1);
  Mat XtX = orc::AtB(X, X);
  Mat I(d, d);
  for (int i = 0; i < d; ++i) I(i, i) = 1.0;
  Mat XtXinv;
  if (!orc::chol_solve(XtX, I, &XtXinv)) return -1;
  Mat Xty = orc::AtB(X, Y);
  Mat B = orc::mul(XtXinv, Xty);
  double rss = 0;
  for (int64_t i = 0; i < N; ++i) {
    double p = 0;
    for (int j = 0; j < d; ++j) p += X(i, j) * B(j, 0);
    if (pred) pred[i] = p;
    const double r = y[i] - p;
    if (resid) resid[i] = r;
    rss += r * r;
  }
  for (int j = 0; j < d; ++j) beta[j] = B(j, 0);
  *sigma2 = rss / (double)N;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// LogisticRegression::FitLogisticModel(X, y, nrrounds)  (regression/LogisticRegression.cpp:279-336)
// Note p and V are those computed at the top of the last executed round, i.e. BEFORE the final
// beta update (as in the reference: Eigen_to_G(w->p), Eigen_to_G(w->V) after the loop).
// ---------------------------------------------------------------------------------------------
int orc_fit_logistic(const double* Xp, const double* y, int64_t N, int d, int nrrounds, double* beta_out,
                     double* p_out, double* v_out) {
  Mat X = wrap(Xp, N, d);
  std::vector<double> beta(d, 0.0), eta(N), p(N), V(N);
  int rounds = 0;
  double lastDeviance = -99999, currentDeviance = -9999;
  while (rounds < nrrounds) {
    for (int64_t i = 0; i < N; ++i) {
      double e = 0;
      for (int j = 0; j < d; ++j) e += X(i, j) * beta[j];
      eta[i] = e;
      p[i] = 1.0 / (1.0 + std::exp(-e));
      V[i] = p[i] * (1.0 - p[i]);
    }
    Mat D = orc::AtB(X, X, V.data());
    Mat r(d, 1);
    for (int j = 0; j < d; ++j) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s += X(i, j) * (y[i] - p[i]);
      r(j, 0) = s;
    }
    Mat delta;
    if (!orc::chol_solve(D, r, &delta)) return -1;
    for (int j = 0; j < d; ++j) beta[j] += delta(j, 0);
    // GetDeviance() (LogisticRegression.cpp:75-94) uses the stored p (pre-update)
    double ll = 0.0;
    for (int64_t i = 0; i < N; ++i) ll += y[i] * std::log(p[i]) + (1. - y[i]) * std::log(1.0 - p[i]);
    currentDeviance = -2.0 * ll;
    if (rounds > 1 && std::fabs(currentDeviance - lastDeviance) < 1e-3) {
      rounds = 0;
      break;
    }
    if (std::fpclassify(currentDeviance) != FP_NORMAL) return -1;
    lastDeviance = currentDeviance;
    rounds++;
  }
  if (rounds == nrrounds) return -1;
  for (int j = 0; j < d; ++j) beta_out[j] = beta[j];
  for (int64_t i = 0; i < N; ++i) {
    p_out[i] = p[i];
    v_out[i] = V[i];
  }
  return 0;
}

// DataConsolidator::imputeGenotypeToMean (src/DataConsolidator.cpp:217-245); the counter check
// "getNumMissing() == 0 -> skip" is equivalent to "no negative entry in the column".
void orc_impute_mean(double* Gp, int64_t N, int M) {
  for (int j = 0; j < M; ++j) {
    double* g = Gp + (size_t)j * N;
    bool any = false;
    for (int64_t i = 0; i < N; ++i)
      if (g[i] < 0) {
        any = true;
        break;
      }
    if (!any) continue;
    int ac = 0, an = 0;
    for (int64_t i = 0; i < N; ++i)
      if (g[i] >= 0) {
        ac += g[i];  // int += double: truncates the running sum, as in the reference (quirk #5)
        an += 2;
      }
    const double p = (an == 0) ? 0.0 : 1.0 * ac / an;
    const double v = 2.0 * p;
    for (int64_t i = 0; i < N; ++i)
      if (g[i] < 0) g[i] = v;
  }
}

// GenotypeCounter::add / getAF (src/GenotypeCounter.h:14-51)
void orc_counter_af(const double* Graw, int64_t N, int M, double* af) {
  for (int j = 0; j < M; ++j) {
    const double* g = Graw + (size_t)j * N;
    double sumAC = 0;
    int64_t nSample = 0;
    for (int64_t i = 0; i < N; ++i) {
      const double x = g[i];
      if (x < 0) {
      } else if (x < 2.0 / 3) {
        sumAC += x;
      } else if (x < 4.0 / 3) {
        sumAC += x;
      } else if (x <= 2.0) {
        sumAC += x;
      }
      ++nSample;
    }
    af[j] = nSample ? 0.5 * sumAC / nSample : -1.0;
  }
}

int orc_flip_poly(const double* Gp, int64_t N, int M, double* out, int* flipped, int* kept) {
  Mat G = wrap(Gp, N, M);
  std::vector<int> f, k;
  Mat O = flip_poly(G, &f, &k);
  if (out) std::memcpy(out, O.a.data(), sizeof(double) * O.a.size());
  for (int j = 0; j < M; ++j) {
    if (flipped) flipped[j] = f[j];
    if (kept) kept[j] = k[j];
  }
  return (int)O.c;
}

// cmcCollapse / zegginiCollapse (src/Model.cpp:73-89, 115-130) on an already flipped/filtered block
void orc_collapse(const double* Gp, int64_t N, int M, int which, double* out) {
  for (int64_t p = 0; p < N; ++p) {
    double o = 0.0;
    for (int m = 0; m < M; ++m) {
      const int g = (int)(Gp[(size_t)m * N + p]);
      if (g > 0) {
        if (which == 0) {
          o = 1.0;
          break;
        }
        o += 1.0;
      }
    }
    out[p] = o;
  }
}

// ---------------------------------------------------------------------------------------------
// SkatTest::fit (src/Model.h:2630-2720) + Skat::SkatImpl::Fit (regression/Skat.cpp:29-105), fp64,
// with P0 folded:  K_sqrt P0 K_sqrt' = W½ (G'VG − G'VX (X'VX)^-1 X'VG) W½      
// ---------------------------------------------------------------------------------------------
int orc_skat(const double* Gp, const double* af, int64_t N, int M, const double* Xp, int d, const double* res,
             const double* v, int binary, double beta1, double beta2, orc_kernel_result* out) {
  (void)binary;
  std::memset(out, 0, sizeof(*out));
  Mat G0 = wrap(Gp, N, M);
  std::vector<int> fl, kp;
  Mat G = flip_poly(G0, &fl, &kp);
  const int m = (int)G.c;
  out->n_poly = m;
  if (m == 0) return -1;
  std::vector<double> w = beta_weights(af, m, beta1, beta2, true);
  std::vector<double> wsq(m);
  for (int i = 0; i < m; ++i) wsq[i] = std::sqrt(w[i]);
  Mat X = wrap(Xp, N, d);
  // Q = || W½ G' r ||²
  double Q = 0;
  for (int j = 0; j < m; ++j) {
    double s = 0;
    const double* g = G.col(j);
    for (int64_t i = 0; i < N; ++i) s += g[i] * res[i];
    s *= wsq[j];
    Q += s * s;
  }
  Mat S = orc::AtB(G, G, v), T = orc::AtB(G, X, v), C = orc::AtB(X, X, v);
  Mat Tt = orc::transpose(T), CiTt;
  if (!orc::sym_solve(C, Tt, &CiTt)) return -1;
  Mat TCT = orc::mul(T, CiTt);
  Mat K(m, m);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) K(i, j) = wsq[i] * (S(i, j) - TCT(i, j)) * wsq[j];
  std::vector<double> ev = orc::sym_eigvals(K);
  out->Q = Q;
  skat_pvalue(ev, N, m, Q, out);
  out->fit_ok = 1;
  return 0;
}

// Literal Skat::SkatImpl::Fit (regression/Skat.cpp:29-105) with the N x N P0 — small N only.
int orc_skat_literal(const double* Gp, const double* af, int64_t N, int M, const double* Xp, int d,
                     const double* res, const double* v, int binary, double beta1, double beta2, int use_float,
                     orc_kernel_result* out) {
  (void)binary;
  std::memset(out, 0, sizeof(*out));
  Mat G0 = wrap(Gp, N, M);
  std::vector<int> fl, kp;
  Mat G = flip_poly(G0, &fl, &kp);
  const int m = (int)G.c;
  out->n_poly = m;
  if (m == 0) return -1;
  std::vector<double> w = beta_weights(af, m, beta1, beta2, true);
  auto F = [&](double x) { return use_float ? (double)(float)x : x; };
  // K_sqrt = diag(sqrt w) G'   (m x N)
  Mat Ks(m, N);
  for (int j = 0; j < m; ++j) {
    const double ws = use_float ? (double)std::sqrt((float)w[j]) : std::sqrt(w[j]);
    for (int64_t i = 0; i < N; ++i) Ks(j, i) = F(ws * F(G(i, j)));
  }
  double Q = 0;
  for (int j = 0; j < m; ++j) {
    double s = 0;
    for (int64_t i = 0; i < N; ++i) s = F(s + F(Ks(j, i) * F(res[i])));
    Q = F(Q + F(s * s));
  }
  Mat P0(N, N);
  if (d == 1) {
    double vs = 0;
    for (int64_t i = 0; i < N; ++i) vs = F(vs + F(v[i]));
    for (int64_t i = 0; i < N; ++i)
      for (int64_t j = 0; j < N; ++j) P0(i, j) = F(-F(v[i]) * F(v[j]) / vs);
    for (int64_t i = 0; i < N; ++i) P0(i, i) = F(P0(i, i) + F(v[i]));
  } else {
    Mat X = wrap(Xp, N, d);
    Mat XtV(d, N);
    for (int k = 0; k < d; ++k)
      for (int64_t i = 0; i < N; ++i) XtV(k, i) = F(F(X(i, k)) * F(v[i]));
    Mat XtVX = orc::mul(XtV, X);
    Mat I(d, d);
    for (int i = 0; i < d; ++i) I(i, i) = 1;
    Mat inv;
    if (!orc::sym_solve(XtVX, I, &inv)) return -1;
    Mat t = orc::mul(inv, XtV);  // d x N
    for (int64_t i = 0; i < N; ++i)
      for (int64_t j = 0; j < N; ++j) {
        double s = 0;
        for (int k = 0; k < d; ++k) s += XtV(k, i) * t(k, j);
        P0(i, j) = F(-s);
      }
    for (int64_t i = 0; i < N; ++i) P0(i, i) = F(P0(i, i) + F(v[i]));
  }
  Mat KP = orc::mul(Ks, P0);  // m x N
  Mat K(m, m);
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(KP(a, i) * Ks(b, i)));
      K(a, b) = s;
    }
  std::vector<double> ev = orc::sym_eigvals(K);
  out->Q = Q;
  skat_pvalue(ev, N, m, Q, out);
  out->fit_ok = 1;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// SkatOTest::fit (src/Model.h:2787-2860) + SkatO::SkatOImpl::Fit / FitSKAT (regression/SkatO.cpp:60-281),
// literal operation order.
// ---------------------------------------------------------------------------------------------
int orc_skato(const double* Gp, const double* af, int64_t N, int M, const double* Xp, int d, const double* res,
              const double* v, int binary, double beta1, double beta2, orc_kernel_result* out) {
  std::memset(out, 0, sizeof(*out));
  Mat G0 = wrap(Gp, N, M);
  std::vector<int> fl, kp;
  Mat G = flip_poly(G0, &fl, &kp);
  const int m = (int)G.c;
  out->n_poly = m;
  if (m == 0) return -1;
  std::vector<double> w = beta_weights(af, m, beta1, beta2, false);
  Mat X = wrap(Xp, N, d);
  // G = G * diag(w)
  for (int j = 0; j < m; ++j) {
    double* g = G.col(j);
    for (int64_t i = 0; i < N; ++i) g[i] *= w[j];
  }
  double rss = 0;
  for (int64_t i = 0; i < N; ++i) rss += res[i] * res[i];

  if (m == 1) {  // FitSKAT, SkatO.cpp:60-99
    out->rho = 0;
    double t = 0;
    for (int64_t i = 0; i < N; ++i) t += res[i] * G(i, 0);
    double Q = t * t;
    if (!binary) Q /= rss / (double)(N - 1);
    Q /= 2.;
    Mat W;
    if (!binary) {
      Mat GtG = orc::AtB(G, G), GtX = orc::AtB(G, X), XtX = orc::AtB(X, X), XtG = orc::transpose(GtX), sol;
      if (!orc::sym_solve(XtX, XtG, &sol)) return -1;
      Mat c = orc::mul(GtX, sol);
      W = GtG;
      W(0, 0) -= c(0, 0);
    } else {
      Mat GtG = orc::AtB(G, G, v), GtX = orc::AtB(G, X, v), XtX = orc::AtB(X, X, v), XtG = orc::transpose(GtX),
          sol;
      if (!orc::sym_solve(XtX, XtG, &sol)) return -1;
      Mat c = orc::mul(GtX, sol);
      W = GtG;
      W(0, 0) -= c(0, 0);
    }
    W(0, 0) /= 2;
    std::vector<double> lam;
    if (get_eigen(W, &lam)) return -1;
    out->Q = Q;
    out->n_lambda = (int)lam.size();
    for (size_t i = 0; i < lam.size() && i < 512; ++i) out->lambda[i] = lam[i];
    out->pvalue = pval_davies(Q, lam);
    out->fit_ok = 1;
    return 0;
  }

  SkatOState st;
  st.nRho = 11;
  double rhosOriginal[11];
  for (int i = 0; i <= 10; ++i) {
    rhosOriginal[i] = 1.0 * i / 10;
    st.rhos[i] = rhosOriginal[i] > 0.999 ? 0.999 : rhosOriginal[i];
  }
  double s2;
  if (binary)
    s2 = 1;
  else {
    s2 = std::sqrt(rss);
    s2 = (s2 * s2) / (double)(N - 1);
  }
  // Qs
  std::vector<double> u(m);
  for (int j = 0; j < m; ++j) {
    double s = 0;
    const double* g = G.col(j);
    for (int64_t i = 0; i < N; ++i) s += res[i] * g[i];
    u[j] = s;
  }
  double Qs[11];
  for (int i = 0; i < 11; ++i) {
    // v * R_rho * v'
    double q = 0;
    for (int a = 0; a < m; ++a) {
      double t = 0;
      for (int b = 0; b < m; ++b) t += u[b] * (a == b ? 1.0 : st.rhos[i]);
      q += t * u[a];
    }
    q /= s2;
    q /= 2.0;
    Qs[i] = q;
    out->Qs[i] = q;
  }
  // Z1
  Mat Z1(N, m);
  if (!binary) {
    Mat XtX = orc::AtB(X, X), XtG = orc::AtB(X, G), sol;
    if (!orc::sym_solve(XtX, XtG, &sol)) return -1;
    Mat XS = orc::mul(X, sol);
    for (size_t k = 0; k < Z1.a.size(); ++k) Z1.a[k] = G.a[k] - XS.a[k];
  } else {
    Mat XtVX = orc::AtB(X, X, v), XtVG = orc::AtB(X, G, v), sol;
    if (!orc::sym_solve(XtVX, XtVG, &sol)) return -1;
    Mat XS = orc::mul(X, sol);
    for (int j = 0; j < m; ++j)
      for (int64_t i = 0; i < N; ++i) {
        const double vs = std::sqrt(v[i]);
        Z1(i, j) = vs * G(i, j) - vs * XS(i, j);
      }
  }
  const double rt2 = std::sqrt(2);
  for (double& z : Z1.a) z = z / rt2;

  std::vector<std::vector<double>> lambdas(11);
  for (int i = 0; i < 11; ++i) {
    Mat R(m, m);
    for (int a = 0; a < m; ++a)
      for (int b = 0; b < m; ++b) R(a, b) = (a == b) ? 1.0 : st.rhos[i];
    Mat L;
    if (!orc::cholesky(R, &L)) return -1;
    Mat Z2 = orc::mul(Z1, L);
    Mat K = orc::AtB(Z2, Z2);
    if (get_eigen(K, &lambdas[i])) return -1;
  }
  // z_bar etc.  SkatO.cpp:178-195
  std::vector<double> zbar(N, 0.0);
  for (int j = 0; j < m; ++j) {
    const double* z = Z1.col(j);
    for (int64_t i = 0; i < N; ++i) zbar[i] += z[i];
  }
  double z_norm = 0;
  for (int64_t i = 0; i < N; ++i) {
    zbar[i] /= (double)m;
    z_norm += zbar[i] * zbar[i];
  }
  std::vector<double> zz(m);  // z_bar' * Z1
  for (int j = 0; j < m; ++j) {
    double s = 0;
    const double* z = Z1.col(j);
    for (int64_t i = 0; i < N; ++i) s += zbar[i] * z[i];
    zz[j] = s;
  }
  Mat ZMZ(m, m), ZIMZ = orc::AtB(Z1, Z1);
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) {
      ZMZ(a, b) = zz[a] * zz[b] / z_norm;
      ZIMZ(a, b) -= ZMZ(a, b);
    }
  if (get_eigen(ZIMZ, &st.lambda)) return -1;
  double varZeta = 0;
  for (size_t k = 0; k < ZMZ.a.size(); ++k) varZeta += ZMZ.a[k] * ZIMZ.a[k];
  st.VarZeta = 4.0 * varZeta;
  double lsum = 0, l2 = 0, l4 = 0;
  for (double l : st.lambda) {
    lsum += l;
    l2 += l * l;
    l4 += l * l * l * l;
  }
  st.lambda_sum = lsum;
  st.MuQ = lsum;
  st.VarQ = 2.0 * l2 + st.VarZeta;
  const double KerQ = l4 / l2 / l2 * 12;
  st.Df = 12 / KerQ;
  // taus  SkatO.cpp:198-203
  double zzsq = 0;
  for (int j = 0; j < m; ++j) zzsq += zz[j] * zz[j];
  for (int i = 0; i < 11; ++i) {
    st.taus[i] = (double)(m * m) * st.rhos[i] * z_norm + (1.0 - st.rhos[i]) * zzsq / z_norm;
    out->taus[i] = st.taus[i];
  }
  // moments, p per rho
  Moment mom[11];
  double pvals[11];
  for (int i = 0; i < 11; ++i) {
    get_moment(lambdas[i], &mom[i]);
    const double Q_Norm = (Qs[i] - mom[i].muQ) / std::sqrt(mom[i].varQ) * std::sqrt(2. * mom[i].df) + mom[i].df;
    pvals[i] = orc_chisq_Q(Q_Norm, mom[i].df);
    out->pvals[i] = pvals[i];
  }
  double minP = pvals[0];
  int minIndex = 0;
  for (int i = 1; i < 11; ++i)
    if (pvals[i] < minP) {
      minP = pvals[i];
      minIndex = i;
    }
  double rho = st.rhos[minIndex];
  out->Q = Qs[minIndex];
  out->minP = minP;
  for (int i = 0; i < 11; ++i) {
    const double q_org = orc_chisq_Qinv(minP, mom[i].df);
    st.Qs_minP[i] = (q_org - mom[i].df) / std::sqrt(2. * mom[i].df) * std::sqrt(mom[i].varQ) + mom[i].muQ;
    out->qminp[i] = st.Qs_minP[i];
  }
  out->muQ = st.MuQ;
  out->varQ = st.VarQ;
  out->varZeta = st.VarZeta;
  out->df = st.Df;
  out->n_lambda = (int)st.lambda.size();
  for (size_t i = 0; i < st.lambda.size() && i < 512; ++i) out->lambda[i] = st.lambda[i];
  // integrate  SkatO.cpp:236-256
  double result = 0, abserr = 0;
  int neval = 0;
  int status = orc_qags(integrand_davies, &st, 0., 40., 1e-25, 0.0001220703, 1000, &result, &abserr, &neval);
  out->qags_status = status;
  out->qags_neval = neval;
  if (status) {
    int n2 = 0;
    int status2 = orc_qags(integrand_liu, &st, 0., 40., 1e-25, 0.0001220703, 1000, &result, &abserr, &n2);
    out->qags_status = status * 100 + status2;
    out->qags_neval += n2;
  }
  double pValue = 1.0 - result;
  // SkatO.cpp:262-277
  const int multi = 3;  // nRho = 11 >= 3
  if (pValue <= 0) {
    const double p = minP * multi;
    if (pValue < p) pValue = p;
  }
  if (pValue == 0.0) {
    pValue = pvals[0];
    for (int i = 1; i < 11; ++i)
      if (pvals[i] > 0 && pvals[i] < pValue) pValue = pvals[i];
  }
  if (rho >= 0.999) rho = 1.;
  out->rho = rho;
  out->pvalue = pValue;
  out->fit_ok = 1;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// CMCTest::fit / ZegginiTest::fit (src/Model.h:821-858, 1177-1215): collapse, REFIT the null model for
// every gene, score test with m = 1 (LinearRegressionScoreTest.cpp:173-263 /
// LogisticRegressionScoreTest.cpp:220-302).  For binary traits with covariates (d > 1) the reference
// evaluates SS(1x1).llt().solve(Identity(d,d)) — an Eigen dimension mismatch;
// the mathematically intended 1-df statistic U²/SS is returned here.
// ---------------------------------------------------------------------------------------------
int orc_burden(const double* Gp, int64_t N, int M, const double* Xp, int d, const double* y, int binary, int which,
               orc_burden_result* out) {
  std::memset(out, 0, sizeof(*out));
  Mat G0 = wrap(Gp, N, M);
  std::vector<int> fl, kp;
  Mat G = flip_poly(G0, &fl, &kp);
  const int m = (int)G.c;
  out->n_poly = m;
  if (m == 0) return -1;
  std::vector<double> c(N);
  orc_collapse(G.a.data(), N, m, which, c.data());
  int nonref = 0;
  for (int64_t i = 0; i < N; ++i) nonref += (c[i] == 0.0) ? 0 : 1;
  out->nonref_site = nonref;
  Mat X = wrap(Xp, N, d);
  std::vector<double> beta(d), pred(N), resid(N), vv(N);
  double sigma2 = 0;
  Mat Cm(N, 1);
  std::memcpy(Cm.a.data(), c.data(), sizeof(double) * (size_t)N);
  double U = 0, SS, stat;
  if (!binary) {
    if (orc_fit_linear(Xp, y, N, d, beta.data(), pred.data(), resid.data(), &sigma2)) return -1;
    for (int64_t i = 0; i < N; ++i) U += c[i] * resid[i];
    Mat ss = orc::AtB(Cm, Cm), SZ = orc::AtB(Cm, X), ZZ = orc::AtB(X, X), I(d, d), ZZi;
    for (int i = 0; i < d; ++i) I(i, i) = 1;
    if (!orc::chol_solve(ZZ, I, &ZZi)) return -1;
    Mat t = orc::mul(orc::mul(SZ, ZZi), orc::transpose(SZ));
    SS = ss(0, 0) - t(0, 0);
    out->V = SS * sigma2;
    // SS = SS.llt().solve(I); beta = SS*U; SS /= sigma2; S = U' SS U
    if (!(SS > 0)) return -1;  // llt of a non-positive 1x1 yields NaN -> stat<0 false -> p NaN; treat as failure
    double SSi = 1.0 / SS;
    SSi /= sigma2;
    stat = U * SSi * U;
  } else {
    if (orc_fit_logistic(Xp, y, N, d, 100, beta.data(), pred.data(), vv.data())) return -1;
    for (int64_t i = 0; i < N; ++i) U += (y[i] - pred[i]) * c[i];
    Mat ss = orc::AtB(Cm, Cm, vv.data()), SZ = orc::AtB(Cm, X, vv.data()), ZZ = orc::AtB(X, X, vv.data()), I(d, d),
        ZZi;
    for (int i = 0; i < d; ++i) I(i, i) = 1;
    if (!orc::chol_solve(ZZ, I, &ZZi)) return -1;
    Mat t = orc::mul(orc::mul(SZ, ZZi), orc::transpose(SZ));
    SS = ss(0, 0) - t(0, 0);
    out->V = SS;
    if (!(SS > 0)) return -1;
    stat = U * (1.0 / SS) * U;
  }
  out->U = U;
  out->stat = stat;
  if (stat < 0) return -1;
  out->pvalue = orc_chisq_Q(stat, 1.0);
  out->fit_ok = 1;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// MetaScoreTest::fit, unrelated samples (src/Model.h:3232-3258): the null model is fitted once
// (MetaUnrelatedQtl::FitNullModel :3503-3515 / MetaUnrelatedBinary::FitNullModel :3672-3705), then every
// genotype column — imputed, NOT flipped — is tested alone; monomorphic sites are skipped (:3246-3250).
//   quantitative: LinearRegressionScoreTest::TestCovariate (LinearRegressionScoreTest.cpp:173-263) and the
//     accessors of MetaUnrelatedQtl (:3543-3549), GetSEBeta (LinearRegressionScoreTest.cpp:365-376)
//   binary: LogisticRegressionScoreTest::TestCovariate (LogisticRegressionScoreTest.cpp:220-302) and the
//     accessors of MetaUnrelatedBinary (:3754-3769); the d > 1 dimension mismatch is SURVEY quirk #15, as in
//     orc_burden.
// covb_diag: what PrintNullModel prints next to beta (LinearRegression.cpp:62-66, LogisticRegression.cpp:330-334).
// ---------------------------------------------------------------------------------------------
int orc_metascore(const double* Gp, int64_t N, int V, const double* Xp, int d, const double* y, int binary, int* ok,
                  double* ustat, double* vstat, double* effect, double* se, double* pval, double* beta_out,
                  double* covb_diag, double* sigma2_out) {
  Mat X = wrap(Xp, N, d);
  std::vector<double> beta(d), pred(N), resid(N), vv(N, 1.0);
  double sigma2 = 1.0;
  if (!binary) {
    if (orc_fit_linear(Xp, y, N, d, beta.data(), pred.data(), resid.data(), &sigma2)) return -1;
  } else {
    if (orc_fit_logistic(Xp, y, N, d, 100, beta.data(), pred.data(), vv.data())) return -1;
    for (int64_t i = 0; i < N; ++i) resid[i] = y[i] - pred[i];
  }
  Mat ZZ = binary ? orc::AtB(X, X, vv.data()) : orc::AtB(X, X), I(d, d), ZZi;
  for (int i = 0; i < d; ++i) I(i, i) = 1;
  if (!orc::chol_solve(ZZ, I, &ZZi)) return -1;
  for (int k = 0; k < d; ++k) {
    if (beta_out) beta_out[k] = beta[k];
    if (covb_diag) covb_diag[k] = ZZi(k, k) * (binary ? 1.0 : sigma2);
  }
  if (sigma2_out) *sigma2_out = sigma2;
  for (int h = 0; h < V; ++h) {
    const double* g = Gp + (size_t)h * N;
    ok[h] = 0;
    ustat[h] = vstat[h] = effect[h] = se[h] = 0.0;
    pval[h] = 1.0;
    bool mono = true;
    for (int64_t i = 1; i < N; ++i)
      if (g[i] != g[0]) {
        mono = false;
        break;
      }
    if (mono) continue;
    Mat Cm(N, 1);
    std::memcpy(Cm.a.data(), g, sizeof(double) * (size_t)N);
    double U = 0;
    for (int64_t i = 0; i < N; ++i) U += g[i] * resid[i];
    Mat ss = binary ? orc::AtB(Cm, Cm, vv.data()) : orc::AtB(Cm, Cm);
    Mat SZ = binary ? orc::AtB(Cm, X, vv.data()) : orc::AtB(Cm, X);
    Mat t = orc::mul(orc::mul(SZ, ZZi), orc::transpose(SZ));
    const double SS = ss(0, 0) - t(0, 0);
    if (!(SS > 0)) continue;
    double stat;
    if (!binary) {
      const double Vlin = SS * sigma2;
      ustat[h] = U / sigma2;
      vstat[h] = Vlin / sigma2 / sigma2;
      effect[h] = (Vlin != 0.0) ? U / SS : 0.0;
      se[h] = sigma2 / std::sqrt(Vlin);
      double SSi = 1.0 / SS;
      SSi /= sigma2;
      stat = U * SSi * U;
    } else {
      ustat[h] = U;
      vstat[h] = SS;
      effect[h] = (U != 0.0) ? U / SS : 0.0;
      se[h] = 1.0 / std::sqrt(SS);
      stat = U * (1.0 / SS) * U;
    }
    if (stat < 0) continue;
    pval[h] = orc_chisq_Q(stat, 1.0);
    ok[h] = 1;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// SKAT adaptive permutation (src/Model.h:2707-2717, src/Permutation.h:69-98, src/LinearAlgebra.h:8-21,
// regression/Skat.cpp:107-116).  The residual vector is permuted cumulatively.
// ---------------------------------------------------------------------------------------------
// The adaptive stop rule of Permutation (src/Permutation.h:69-98): init / next / add / getPvalue.  Used by orc_skat_permute
// below and exported alone (orc_perm_stop_run) so that tests can feed the SAME sequence of statistics to it and to the
// reference's compiled class (oracle/_ref/libref_host.so).
namespace {
struct PermStop {
  int numPerm = 0, actualPerm = 0, numX = 0, numEqual = 0;
  int threshold = 0;  // (an int in the reference, src/Permutation.h:153: the product below is truncated)
  double alpha = 0, obs = 0;
  void init(int nPerm, double a, double observation) {
    numPerm = nPerm;
    alpha = a;
    obs = observation;
    actualPerm = 0;
    threshold = (int)(1.0 * nPerm * a * 2);
    numX = 0;
    numEqual = 0;
  }
  bool next() const {
    if (actualPerm >= numPerm) return false;
    if (numX + numEqual >= threshold) return false;
    return true;
  }
  void add(double s) {
    actualPerm++;
    if (s > obs) numX++;
    if (s == obs) numEqual++;
  }
  double pvalue() const { return actualPerm == 0 ? 1.0 : 1.0 * (numX + 0.5 * numEqual) / actualPerm; }
};
}  // namespace

// feed stats[0 .. n) until the rule stops (or the list ends); out3 = actualPerm, numX, numEqual; returns the p-value
extern "C" double orc_perm_stop_run(int nPerm, double alpha, double obs, const double* stats, int n, int* out3) {
  PermStop p;
  p.init(nPerm, alpha, obs);
  int used = 0;
  while (p.next() && used < n) p.add(stats[used++]);
  out3[0] = p.actualPerm;
  out3[1] = p.numX;
  out3[2] = p.numEqual;
  return p.pvalue();
}

int orc_skat_permute(const double* Gp, const double* af, int64_t N, int M, const double* res, double beta1,
                     double beta2, double obs, int nPerm, double alpha, int use_float, orc_perm_result* out) {
  Mat G0 = wrap(Gp, N, M);
  std::vector<int> fl, kp;
  Mat G = flip_poly(G0, &fl, &kp);
  const int m = (int)G.c;
  if (m == 0) return -1;
  std::vector<double> w = beta_weights(af, m, beta1, beta2, true);
  auto F = [&](double x) { return use_float ? (double)(float)x : x; };
  Mat Ks(m, N);
  for (int j = 0; j < m; ++j) {
    const double ws = use_float ? (double)std::sqrt((float)w[j]) : std::sqrt(w[j]);
    for (int64_t i = 0; i < N; ++i) Ks(j, i) = F(ws * F(G(i, j)));
  }
  std::vector<double> pr(res, res + N);
  PermStop stop;
  stop.init(nPerm, alpha, obs);
  while (stop.next()) {
    for (int64_t i = N - 1; i >= 1; --i) {
      const int64_t j = orc_rand() % (i + 1);
      if (i != j) std::swap(pr[i], pr[j]);
    }
    double Q = 0;
    for (int j = 0; j < m; ++j) {
      double s = 0;
      for (int64_t i = 0; i < N; ++i) s = F(s + F(Ks(j, i) * F(pr[i])));
      Q = F(Q + F(s * s));
    }
    stop.add(Q);
  }
  out->num_perm = nPerm;
  out->actual_perm = stop.actualPerm;
  out->num_x = stop.numX;
  out->num_equal = stop.numEqual;
  out->threshold = stop.threshold;
  out->pvalue = stop.pvalue();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// MetaCovTest, unrelated samples  (src/Model.cpp:844-1004)
//   QT     (MetaCovUnrelatedQtl, :506-593): g <- g - mean(g) in float; covXX = g1.g2 / sigma2;
//          covXZ = g'Z / sigma2 (Z = covariates with intercept, NOT centred); covZZ = Zc'Zc / sigma2 (Zc centred,
//          double); covZZInv = LDLT solve of the identity (the zero pivot of the centred intercept yields a zero
//          row/column, Eigen LDLT::solve)
//   binary (MetaCovUnrelatedBinary, :694-778): no centring; covXX = sum g1 w g2, covXZ = g'WZ, covZZ = Z'WZ with
//          w = p(1-p) stored as float
//   value printed for (head h, marker j): covXX(h,j) - covXZ_h' covZZInv covXZ_j  (computeScaledXX, Model.h:3997-4005),
//   float throughout, scaled by 1/N only when written.
//   Window (Model.h:3956-3990): a new variant evicts queue heads while |pos_new - pos_head| > window or the
//   chromosome differs; monomorphic variants never enter the queue (but still evict).  The row of head h therefore
//   holds h and every later kept variant pushed before h's eviction.
// ---------------------------------------------------------------------------------------------
int orc_metacov(const double* Gp, int64_t N, int V, const int* chrom, const int* pos, const double* Xp,
                const double* y, int d, int binary, int window, int use_float, int* kept, double* cov, int* row_end,
                double* xz, double* zz) {
  Mat G = wrap(Gp, N, V);
  Mat X = wrap(Xp, N, d);
  auto F = [&](double x) { return use_float ? (double)(float)x : x; };
  // ---- null model -----------------------------------------------------------------------------
  double sigma2 = 0;
  std::vector<double> w(N, 1.0);
  if (!binary) {
    std::vector<double> beta(d), pred(N), res(N);
    if (orc_fit_linear(Xp, y, N, d, beta.data(), pred.data(), res.data(), &sigma2)) return -1;
    sigma2 = F(sigma2);  // `float sigma2` member (Model.cpp:592)
  } else {
    std::vector<double> beta(d), p(N), v(N);
    if (orc_fit_logistic(Xp, y, N, d, 100, beta.data(), p.data(), v.data())) return -1;
    for (int64_t i = 0; i < N; ++i) w[i] = F(v[i]);  // Eigen::VectorXf weight
  }
  // ---- covZZ, covZZInv (double) ---------------------------------------------------------------
  Mat ZZ(d, d);
  if (!binary) {
    std::vector<double> mean(d, 0.0);
    for (int k = 0; k < d; ++k) {
      double sx = 0;
      for (int64_t i = 0; i < N; ++i) sx += X(i, k);
      mean[k] = sx / (double)N;
    }
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) {
        double sx = 0;
        for (int64_t i = 0; i < N; ++i) sx += (X(i, a) - mean[a]) * (X(i, b) - mean[b]);
        ZZ(a, b) = sx * (1.0 / sigma2);
      }
  } else {
    for (int a = 0; a < d; ++a)
      for (int b = 0; b <= a; ++b) {
        double sx = 0;
        for (int64_t i = 0; i < N; ++i) sx += X(i, a) * w[i] * X(i, b);
        ZZ(a, b) = ZZ(b, a) = sx;
      }
  }
  // LDLT solve with zero pivots mapped to zero: for the centred QT matrix the intercept row/column is exactly 0
  Mat ZZinv(d, d);
  {
    std::vector<int> live;
    for (int a = 0; a < d; ++a) {
      bool zero = true;
      for (int b = 0; b < d; ++b)
        if (ZZ(a, b) != 0.0) zero = false;
      if (!zero) live.push_back(a);
    }
    const int q = (int)live.size();
    if (q > 0) {
      Mat A(q, q), I(q, q), Ai;
      for (int a = 0; a < q; ++a) {
        I(a, a) = 1.0;
        for (int b = 0; b < q; ++b) A(a, b) = ZZ(live[a], live[b]);
      }
      if (!orc::sym_solve(A, I, &Ai)) return -1;
      for (int a = 0; a < q; ++a)
        for (int b = 0; b < q; ++b) ZZinv(live[a], live[b]) = Ai(a, b);
    }
  }
  for (int a = 0; a < d; ++a)
    for (int b = 0; b < d; ++b) zz[a * d + b] = ZZ(a, b);
  // ---- per-variant transform and covXZ ------------------------------------------------------------
  std::vector<std::vector<double>> gt(V);
  for (int j = 0; j < V; ++j) {
    // isMonomorphicMarker on the imputed column (DataConsolidator.cpp:94-116)
    bool mono = true;
    int64_t first = N;
    for (int64_t i = 0; i < N; ++i)
      if (G(i, j) >= 0) {
        first = i;
        break;
      }
    for (int64_t i = first + 1; i < N; ++i) {
      if (G(i, j) < 0) continue;
      if (G(i, j) != G(first, j)) {
        mono = false;
        break;
      }
    }
    kept[j] = mono ? 0 : 1;
    row_end[j] = -1;
    for (int k = 0; k < d; ++k) xz[(size_t)j * d + k] = NAN;
    if (mono) continue;
    std::vector<double>& g = gt[j];
    g.resize(N);
    for (int64_t i = 0; i < N; ++i) g[i] = F(G(i, j));  // assignGenotype: float copy
    if (!binary) {
      double sx = 0;
      for (int64_t i = 0; i < N; ++i) sx = F(sx + g[i]);
      const double avg = F(sx / (double)N);
      for (int64_t i = 0; i < N; ++i) g[i] = F(g[i] - avg);
    }
    for (int k = 0; k < d; ++k) {
      double sx = 0;
      for (int64_t i = 0; i < N; ++i)
        sx = binary ? F(sx + F(F(g[i] * w[i]) * F(X(i, k)))) : F(sx + F(g[i] * F(X(i, k))));
      xz[(size_t)j * d + k] = binary ? sx : F(sx / sigma2);
    }
  }
  // ---- rows -------------------------------------------------------------------------------------------
  for (size_t i = 0; i < (size_t)V * V; ++i) cov[i] = NAN;
  for (int h = 0; h < V; ++h) {
    if (!kept[h]) continue;
    for (int j = h; j < V; ++j) {
      // the variant at j evicts h (and ends its row) when it is out of the window, kept or not
      if (chrom[j] != chrom[h] || std::abs(pos[j] - pos[h]) > window) break;
      if (!kept[j]) continue;
      double xx = 0;
      for (int64_t i = 0; i < N; ++i)
        xx = binary ? F(xx + F(F(gt[h][i] * w[i]) * gt[j][i])) : F(xx + F(gt[h][i] * gt[j][i]));
      if (!binary) xx = F(xx / sigma2);
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

// built-in integrands for the QAGS fixture checks
static double builtin_f(double x, void* p) {
  const double* q = (const double*)p;
  const int id = (int)q[0];
  const double alpha = q[1];
  switch (id) {
    case 0: return std::pow(x, alpha) * std::log(1 / x);           // QUADPACK book f1
    case 1: return std::exp(-x) * std::sin(alpha * x);
    case 2: return orc_chisq_pdf(x, 1.0) * std::exp(-alpha * x);     // x^-1/2 singularity at 0
    case 3: return 1.0 / (1.0 + alpha * x * x);
    case 4: return (x > 0 ? std::pow(x, -0.5) : 0.0) * std::cos(alpha * x);
    default: return 0.0;
  }
}
int orc_qags_builtin(int id, double alpha, double a, double b, double epsabs, double epsrel, int limit,
                     double* result, double* abserr, int* neval_out) {
  double q[2] = {(double)id, alpha};
  return orc_qags(builtin_f, q, a, b, epsabs, epsrel, limit, result, abserr, neval_out);
}

}  // extern "C"
