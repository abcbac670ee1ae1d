// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_api.h).  CPU restatement of the KBAC test as rvtests runs it:
//   KBACTest::fit                      src/Model.h:2925-2998  (quiet, mafUpper = 1, sided = 1: the case model only)
//   KbacTest::KbacTest / calcKbacP / m_checkAdaptivePvalue / m_trimXdat      regression/kbac.cpp:16-391
//   std::random_shuffle (libstdc++: i = 1 .. n-1, j = rand() % (i + 1), swap) on the process-wide rand() stream
// and of the GSL 1.16 routines it calls (the library the reference vendors as third/gsl-1.16.tar.gz):
//   gsl_cdf_hypergeometric_P (cdf/hypergeometric.c), gsl_ran_hypergeometric_pdf (randist/hyperg.c),
//   gsl_sf_lnchoose / gsl_sf_lnfact (specfunc/gamma.c: log of the exact factorial table up to 170!, Lanczos beyond).
// Pinned by tests/golden/kbac.json: p-values, stream positions and hypergeometric values produced by the REFERENCE's own
// kbac.cpp linked against GSL 1.16 built from that tarball (tests/golden/make_kbac_golden.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <list>
#include <string>
#include <vector>

// the oracle's own statement of glibc's rand() (orc_models.cpp; pinned against libc in tests/test_rand_stream.py)
extern "C" int orc_rand(void);

namespace {
// n! for n <= 170 as GSL's table holds it: the exact integer rounded to the nearest double
double exact_factorial(unsigned n) {
  static std::vector<double> table;
  if (table.empty()) {
    std::vector<uint32_t> big(1, 1);  // little-endian base 1e9
    table.push_back(1.0);
    for (unsigned k = 1; k <= 170; ++k) {
      uint64_t carry = 0;
      for (auto& limb : big) {
        const uint64_t v = (uint64_t)limb * k + carry;
        limb = (uint32_t)(v % 1000000000ull);
        carry = v / 1000000000ull;
      }
      while (carry) {
        big.push_back((uint32_t)(carry % 1000000000ull));
        carry /= 1000000000ull;
      }
      std::string s = std::to_string(big.back());
      for (int i = (int)big.size() - 2; i >= 0; --i) {
        std::string t = std::to_string(big[i]);
        s += std::string(9 - t.size(), '0') + t;
      }
      table.push_back(strtod(s.c_str(), nullptr));
    }
  }
  return table[n];
}

double lngamma_lanczos(double x) {  // specfunc/gamma.c lngamma_lanczos: g = 7, 9 coefficients
  static const double c[9] = {0.99999999999980993227684700473478, 676.520368121885098567009190444019,
                              -1259.13921672240287047156078755283, 771.3234287776530788486528258894,
                              -176.61502916214059906584551354,     12.507343278686904814458936853,
                              -0.13857109526572011689554707,       9.984369578019570859563e-6,
                              1.50563273514931155834e-7};
  x -= 1.0;
  double Ag = c[0];
  for (int k = 1; k <= 8; k++) Ag += c[k] / (x + k);
  const double term1 = (x + 0.5) * std::log((x + 7.5) / M_E);
  const double term2 = 0.9189385332046727418 + std::log(Ag);
  return term1 + (term2 - 7.0);
}
double lnfact(unsigned n) { return n <= 170 ? std::log(exact_factorial(n)) : lngamma_lanczos(n + 1.0); }
double lnchoose(unsigned n, unsigned m) {
  if (m == n || m == 0) return 0.0;
  if (m * 2 > n) m = n - m;
  return lnfact(n) - lnfact(m) - lnfact(n - m);
}
double hyper_pdf(unsigned k, unsigned n1, unsigned n2, unsigned t) {
  if (t > n1 + n2) t = n1 + n2;
  if (k > n1 || k > t) return 0;
  if (t > n2 && k + n2 < t) return 0;
  const double c1 = lnchoose(n1, k), c2 = lnchoose(n2, t - k), c3 = lnchoose(n1 + n2, t);
  return std::exp(c1 + c2 - c3);
}
double hyper_P(unsigned k, unsigned n1, unsigned n2, unsigned t) {
  if (k >= n1 || k >= t) return 1.0;
  const double midpoint = ((double)t * n1) / ((double)n1 + (double)n2);
  const double eps = 2.2204460492503131e-16;
  if (k >= midpoint) {  // 1 - upper_tail
    unsigned i = k + 1;
    double s = hyper_pdf(i, n1, n2, t), Q = s;
    while (i < t) {
      const double factor = ((n1 - i) / (i + 1.0)) * ((t - i) / (n2 + i + 1.0 - t));
      s *= factor;
      Q += s;
      if (s / Q < eps) break;
      i++;
    }
    return 1 - Q;
  }
  int i = (int)k;
  double s = hyper_pdf(i, n1, n2, t), P = s;
  while (i > 0) {
    const double factor = (i / (n1 - i + 1.0)) * ((unsigned)(n2 + i - t) / (t - i + 1.0));
    s *= factor;
    P += s;
    if (s / P < eps) break;
    i--;
  }
  return P;
}
}  // namespace

extern "C" {

double orc_hypergeometric_P(unsigned k, unsigned n1, unsigned n2, unsigned t) { return hyper_P(k, n1, n2, t); }

// x: people-major N x M genotypes (flipped-to-minor, polymorphic, imputed); y: 0/1; maf: M frequencies.  Draws from
// the oracle's rand() stream (orc_rand_seed first).  Returns 0; *pvalue as KBACTest prints it; diagnostics optional.
int orc_kbac(const double* x, const double* yin, const double* mafs, int N, int M, int nperm, double alpha, double* pvalue,
             double* obs_out, int* npattern_out, int* perms_done_out) {
  const unsigned adaptive = alpha >= 1.0 ? 0 : 5000;
  std::vector<double> y(yin, yin + N);
  // constructor: invalid codings -> wild type; m_trimXdat: keep 0 < maf <= 1
  std::vector<int> use;
  for (int j = 0; j < M; ++j)
    if (!(mafs[j] <= 0.0 || mafs[j] > 1.0)) use.push_back(j);
  const unsigned regionLen = (unsigned)use.size();
  unsigned nCases = 0;
  for (int i = 0; i < N; ++i)
    if (y[i] == 1.0) ++nCases;
  const unsigned nCtrls = N - nCases;
  std::vector<double> gid(N);
  for (int i = 0; i < N; ++i) {
    double L = 0.0, R = 0.0;
    const double ixiix = std::pow(9.0, 10.0);
    unsigned lastCnt = 0, tmpCnt = 0;
    for (unsigned j = 0; j != regionLen; ++j) {
      double g = x[(size_t)i * M + use[j]];
      if (g != 0.0 && g != 1.0 && g != 2.0) g = 0.0;
      if (g != -9.0 && g != 0.0)
        R += std::pow(3.0, 1.0 * (j - lastCnt)) * g;
      else
        continue;
      if (R >= ixiix) {
        L = L + 1.0;
        R = R - ixiix;
        lastCnt = lastCnt + tmpCnt + 1;
        tmpCnt = 0;
      } else {
        ++tmpCnt;
      }
    }
    gid[i] = L + R * 1e-10;
  }
  std::list<double> uniq(gid.begin(), gid.end());
  uniq.remove(0.0);
  if (uniq.empty()) {
    *pvalue = 1.0;
    if (npattern_out) *npattern_out = 0;
    if (perms_done_out) *perms_done_out = 0;
    return 0;
  }
  uniq.sort();
  uniq.unique();
  const std::vector<double> pat(uniq.begin(), uniq.end());
  const size_t P = pat.size();
  if (npattern_out) *npattern_out = (int)P;
  std::vector<unsigned> cnt(P, 0);
  for (int i = 0; i < N; ++i)
    for (size_t u = 0; u < P; ++u)
      if (gid[i] == pat[u]) {
        ++cnt[u];
        break;
      }
  unsigned iPerm = 0, pc1 = 0, pc2 = 0;
  double observed = 0.0;
  *pvalue = 9.0;
  while (iPerm <= (unsigned)nperm) {
    std::vector<unsigned> sub(P, 0);
    for (int i = 0; i < N; ++i)
      if (y[i] == 1.0)
        for (size_t u = 0; u < P; ++u)
          if (gid[i] == pat[u]) {
            ++sub[u];
            break;
          }
    double kbac = 0.0;
    for (size_t u = 0; u < P; ++u) {
      const double w = hyper_P(sub[u], cnt[u], N - cnt[u], nCases);
      kbac = kbac + ((1.0 * sub[u]) / (1.0 * nCases) - (1.0 * (cnt[u] - sub[u])) / (1.0 * nCtrls)) * w;
    }
    const double statistic = kbac;
    if (iPerm == 0)
      observed = statistic;
    else {
      if (statistic >= observed) ++pc1;
      if (statistic <= observed) ++pc2;
      if (adaptive != 0 && iPerm % adaptive == 0) {  // m_checkAdaptivePvalue(…, alternative = 0)
        const double ap = (1.0 * pc1 + 1.0) / (1.0 * iPerm + 1.0);
        const double sd = std::sqrt(ap * (1.0 - ap) / (1.0 * iPerm));
        if (ap - 6.0 * sd > alpha) *pvalue = ap;
      }
    }
    if (*pvalue <= 1.0) break;
    for (int i = 1; i < N; ++i) {  // std::random_shuffle
      const int j = orc_rand() % (i + 1);
      if (i != j) std::swap(y[i], y[j]);
    }
    ++iPerm;
  }
  if (!(*pvalue <= 1.0)) *pvalue = (1.0 * pc1 + 1.0) / (1.0 * nperm + 1.0);
  if (obs_out) *obs_out = observed;
  if (perms_done_out) *perms_done_out = (int)iPerm;
  return 0;
}

}  // extern "C"
