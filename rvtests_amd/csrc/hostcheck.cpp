// rvtests_amd — HOST TEST HARNESS (test-only; never linked into librvtests_amd.so).
//
// Compiles the RVT_HD device algorithms (special functions, Davies / Liu, QAGS state machine, per-gene
// statistics with flip algebra and the tridiagonal eigen solver, p-value stage) with g++ so that the
// CPU test-suite (`pytest -m "not gpu"`) can exercise them against the oracle without a GPU.
// It consumes sufficient statistics the TEST computes (numpy) — it is not a CPU implementation of the
// engine and the shipped library has no path into it.
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>
#include <chrono>
#include "host_stage.h"
#include "band_tiles.h"
#include "perm_counter.h"
#include "rvt_pvalue.h"
#include "rvt_mvn.h"

using namespace rvt;

#ifdef RVT_DV_PROFILE
// profiling build (tools/davies_divergence.py): what every abscissa of a SKAT-O quadrature spent in qf()'s searches
namespace rvt {
long long rvt_dv_profile[8];
std::vector<long long> g_dv_log;
void rvt_dv_eval_mark() {
  for (int k = 0; k < 5; ++k) g_dv_log.push_back(rvt_dv_profile[k]);
}
std::set<unsigned long long> g_dv_keys[2];
long long g_dv_key_calls[2];
void rvt_dv_key(int kind, double u) {
  unsigned long long b;
  std::memcpy(&b, &u, 8);
  g_dv_keys[kind].insert(b);
  ++g_dv_key_calls[kind];
}
}  // namespace rvt
#endif

extern "C" {

double hc_chisq_Q(double x, double nu) { return chisq_Q(x, nu); }
double hc_chisq_P(double x, double nu) { return chisq_P(x, nu); }
double hc_chisq_Qinv(double q, double nu) { return chisq_quantile_Q(q, nu); }
double hc_chisq_pdf(double x, double nu) { return chisq_density(x, nu); }
double hc_beta_pdf(double x, double a, double b) { return beta_density(x, a, b); }

// term-by-term evaluation in the reference's order (bit-identical to qfc.c)
double hc_davies_pvalue(const double* lam, int n, double Q, int* fault, double* nterms) {
  std::vector<int> th(n > 0 ? n : 1);
  davies_order(lam, n, th.data());
  return davies_pvalue(lam, th.data(), n, Q, fault, nterms, nullptr, false);
}
// same, replaying the c-independent searches from davies_prelude() (what the SKAT-O integrand does)
double hc_davies_pvalue_cached(const double* lam, int n, double Q, int* fault, double* nterms) {
  std::vector<int> th(n > 0 ? n : 1);
  davies_order(lam, n, th.data());
  DaviesPrelude pre;
  davies_prelude(lam, th.data(), n, 10000, 0.000001, &pre, false);
  return davies_pvalue(lam, th.data(), n, Q, fault, nterms, &pre, false);
}
// product form of the coefficient sums (the engine's default), with or without the cached searches
double hc_davies_pvalue_fast(const double* lam, int n, double Q, int cached, int* fault, double* nterms) {
  std::vector<int> th(n > 0 ? n : 1);
  davies_order(lam, n, th.data());
  if (!cached) return davies_pvalue(lam, th.data(), n, Q, fault, nterms, nullptr, true);
  DaviesPrelude pre;
  davies_prelude(lam, th.data(), n, 10000, 0.000001, &pre, true);
  return davies_pvalue(lam, th.data(), n, Q, fault, nterms, &pre, true);
}
// the same against ONE memo of the searches' coefficient sums shared by all the points (what a SKAT-O quadrature does);
// slots[2] = occupied memo slots (errbd, truncation) afterwards
void hc_davies_memo_sweep(const double* lam, int n, const double* Q, int nq, double* p, int* slots) {
  std::vector<int> th(n > 0 ? n : 1);
  davies_order(lam, n, th.data());
  DaviesPrelude pre;
  DaviesMemo memo;
  dv_memo_clear((dv_memo_p)&memo);
  davies_prelude(lam, th.data(), n, 10000, 0.000001, &pre, true, &memo);
  for (int i = 0; i < nq; ++i) {
    int fault;
    double nt;
    p[i] = davies_pvalue(lam, th.data(), n, Q[i], &fault, &nt, &pre, true);
  }
  slots[0] = slots[1] = 0;
  for (int i = 0; i < kMemoSlotsE; ++i) slots[0] += memo.ekey[i] != kMemoEmpty;
  for (int i = 0; i < kMemoSlotsT; ++i) slots[1] += memo.tkey[i] != kMemoEmpty;
}
double hc_liu_pvalue(const double* lam, int n, double Q) { return liu_pvalue(lam, n, Q); }

void hc_sym_eigvals(const double* Ain, int n, double* out) {
  std::vector<double> A(Ain, Ain + (size_t)n * n), d(n), e(n), v(n), w(n), red(64);
  Coop co{0, 1, red.data()};
  coop_sym_eigvals(co, A.data(), n, d.data(), e.data(), v.data(), w.data(), out);
}

// evaluations of the Sturm recurrence per eigenvalue (mean and maximum) of the same solver: what the hybrid of bisection and
// interpolation costs (pure bisection: ~53-62)
void hc_tridiag_evals(const double* din, const double* ein, int n, double* mean_out, int* max_out) {
  std::vector<double> d(din, din + n), e(n, 0.0);
  for (int j = 0; j + 1 < n; ++j) e[j] = ein[j];
  double lo = d[0], hi = d[0];
  for (int j = 0; j < n; ++j) {
    const double r = (j > 0 ? fabs(e[j - 1]) : 0.0) + (j < n - 1 ? fabs(e[j]) : 0.0);
    lo = fmin(lo, d[j] - r);
    hi = fmax(hi, d[j] + r);
  }
  const double span0 = fmax(fabs(lo), fabs(hi));
  int sh = 0;
  if (span0 > 0.0 && span0 < INFINITY) (void)frexp(span0, &sh);
  for (int j = 0; j < n; ++j) {
    d[j] = ldexp(d[j], -sh);
    if (j < n - 1) {
      const double es = ldexp(e[j], -sh);
      e[j] = fmax(es * es, 0x1p-200);
    }
  }
  lo = ldexp(lo, -sh);
  hi = ldexp(hi, -sh);
  const double span = fmax(fabs(lo), fabs(hi)), pivmin = DBL_MIN * 1024.0;
  lo -= 2.0 * kDblEps * span * n + 2.0 * pivmin;
  hi += 2.0 * kDblEps * span * n + 2.0 * pivmin;
  long long tot = 0;
  int mx = 0;
  for (int idx = 0; idx < n; ++idx) {
    int ev = 0;
    (void)sturm_eigenvalue(d.data(), e.data(), n, idx, lo, hi, span, pivmin, &ev);
    tot += ev;
    mx = std::max(mx, ev);
  }
  *mean_out = (double)tot / n;
  *max_out = mx;
}

// the bisection alone: d[n], e[n-1] -> all eigenvalues ascending (division-free Sturm count, rvt_coop.h)
void hc_tridiag_eigvals(const double* din, const double* ein, int n, double* out) {
  std::vector<double> d(din, din + n), e(n, 0.0), red(64);
  for (int j = 0; j + 1 < n; ++j) e[j] = ein[j];
  Coop co{0, 1, red.data()};
  coop_tridiag_eigvals(co, d.data(), e.data(), n, out);
}

#ifdef RVT_DV_PROFILE
// distinct evaluation points u of errbd (kind 0) / truncation (kind 1) since the last call, and the number of evaluations
void hc_dv_keys(long long* out) {
  for (int k = 0; k < 2; ++k) {
    out[2 * k] = (long long)rvt::g_dv_keys[k].size();
    out[2 * k + 1] = rvt::g_dv_key_calls[k];
    rvt::g_dv_keys[k].clear();
    rvt::g_dv_key_calls[k] = 0;
  }
}
int hc_dv_log(long long* out, int cap) {  // 5 cumulative counters per abscissa; returns the number of records and clears
  const int n = (int)(rvt::g_dv_log.size() / 5);
  for (int i = 0; i < n * 5 && i < cap; ++i) out[i] = rvt::g_dv_log[i];
  rvt::g_dv_log.clear();
  return n;
}
#endif

// QAGS state machine driven serially on one of the fixture integrands (ids as in the oracle)
static double builtin_f(int id, double alpha, double x) {
  switch (id) {
    case 0: return pow(x, alpha) * log(1 / x);
    case 1: return exp(-x) * sin(alpha * x);
    case 2: return chisq_density(x, 1.0) * exp(-alpha * x);
    case 3: return 1.0 / (1.0 + alpha * x * x);
    case 4: return (x > 0 ? pow(x, -0.5) : 0.0) * cos(alpha * x);
    default: return 0.0;
  }
}
int hc_qags_builtin(int id, double alpha, double a, double b, double epsabs, double epsrel, int limit,
                    double* result, double* abserr, int* neval) {
  std::vector<char> mem(qags_workspace_bytes(limit));
  QagsWorkspace ws = qags_workspace_carve(mem.data(), limit);
  QagsMachine qm;
  double fv[42];
  int ne = 0;
  qm.begin(a, b, epsabs, epsrel, limit, ws);
  if (qm.running()) {
    for (int t = 0; t < 21; ++t) fv[t] = builtin_f(id, alpha, gk21_abscissa(a, b, t));
    ne += 21;
    qm.first_panel(fv);
  }
  while (qm.running()) {
    double a1, b1, b2;
    qm.bisect(&a1, &b1, &b2);
    for (int t = 0; t < 42; ++t)
      fv[t] = builtin_f(id, alpha, t < 21 ? gk21_abscissa(a1, b1, t) : gk21_abscissa(b1, b2, t - 21));
    ne += 42;
    qm.advance(fv, fv + 21);
  }
  *result = qm.result;
  *abserr = qm.abserr;
  *neval = ne;
  return qm.status;
}

// Full per-gene stage from test-provided sufficient statistics.
//   R: M x (M+d+1) row-major = G' D [G | X | rr];  colstat: 3 x M (sum, min, max);
//   bstats: 2 x (3+d) burden sums (may be null)
int hc_gene(int trait, int64_t N, int d, double sigma2, double rss, double rsum, const double* C,
            const double* Cinv, int M, const double* Rin, const double* colstat_in, const double* bstats,
            const double* af, const rvt_params* prm, unsigned tests, rvt_gene_result* out, int* flip_out,
            int* kept_out, double* lambda_out /* 2*M or null */, double* dbg /* 64 or null */) {
  NullConsts nc;
  std::memset(&nc, 0, sizeof(nc));
  nc.N = N;
  nc.ld = (N + 15) / 16 * 16;
  nc.d = d;
  nc.binary = trait;
  nc.sigma2 = sigma2;
  nc.rss = rss;
  nc.rsum = rsum;
  for (int i = 0; i < d * d; ++i) {
    nc.C[i] = C[i];
    nc.Cinv[i] = Cinv[i];
  }
  const int Cc = M + d + 1;
  const int Mp = (M + 15) / 16 * 16, Cp = (Cc + 15) / 16 * 16;
  std::vector<double> parts((size_t)Mp * Cp, 0.0), cs((size_t)3 * Mp, 0.0);
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < Cc; ++j)
      if ((j >> 4) >= (i >> 4)) parts[(size_t)i * Cp + j] = Rin[(size_t)i * Cc + j];
  for (int k = 0; k < 3; ++k)
    for (int j = 0; j < M; ++j) cs[(size_t)k * Mp + j] = colstat_in[(size_t)k * M + j];
  std::vector<double> mem(gene_scratch_doubles(Mp, Cp) + 16), red(64), lam((size_t)2 * M + 2);
  GeneScratch ws = gene_scratch_carve(mem.data(), Mp, Cp);
  Coop co{0, 1, red.data()};
  GeneStats gs;
  std::memset(&gs, 0, sizeof(gs));
  gene_assemble(co, nc, M, Mp, Cp, parts.data(), 1, cs.data(), bstats, 1, af, *prm, tests, ws, &gs, flip_out,
                kept_out);
  std::vector<double> vec((size_t)8 * Mp + 8);
  for (int w = 0; w < kNTridiag; ++w)
    gene_tridiag(co, nc, w, M, Mp, tests, ws, ws.eig + (size_t)w * Mp * Mp, vec.data(), &gs);
  if (getenv("RVT_SPECTRUM_PER_PROBLEM")) {  // the per-problem form (one workgroup per eigenproblem on the device until round 5)
    for (int k = 0; k < kNEigen; ++k) gene_spectrum(co, nc, k, M, Mp, tests, ws, vec.data(), &gs, lam.data());
  } else {
    std::vector<double> vec2((size_t)39 * Mp + 64);
    SpectrumMeta meta[kNEigen];
    gene_spectrum_all(co, nc, M, Mp, tests, ws, vec2.data(), meta, &gs, lam.data());
  }
  std::vector<int> th1(M + 1), th2(M + 1);
  std::vector<char> qmem(qags_workspace_bytes(kSkatoLimit));
  gene_pvalue_serial(gs, lam.data(), tests, 0, th1.data(), th2.data(), qmem.data(), out);
  if (lambda_out) std::memcpy(lambda_out, lam.data(), sizeof(double) * 2 * M);
  if (dbg) {
    for (int i = 0; i < 11; ++i) {
      dbg[i] = gs.Qs[i];
      dbg[11 + i] = gs.mom_mu[i];
      dbg[22 + i] = gs.mom_var[i];
      dbg[33 + i] = gs.mom_df[i];
      dbg[44 + i] = gs.tau[i];
    }
    dbg[55] = gs.muQ;
    dbg[56] = gs.varQ;
    dbg[57] = gs.varZeta;
    dbg[58] = gs.df;
    dbg[59] = gs.zimz_nlambda;
    dbg[60] = gs.skat_nlambda;
  }
  return 0;
}

// AnalyticVT's band probability exactly as the device evaluates it (rvt_mvn.h): R = correlation (row-major n x n)
double hc_mvn_band(const double* R, int n, double T, double* err) {
  std::vector<double> A(R, R + (size_t)n * n), y(n), alpha(n);
  return mvn_band_prob_serial(A.data(), n, T, y.data(), alpha.data(), err);
}
double hc_mvn_phiinv(double p) { return mvn_phiinv(p); }

// ---- perm_counter.h: the keyed bijection of [0, n) the counter-based SKAT permutations use --------------------------------
void hc_perm_indices(uint64_t seed, uint64_t gene, uint32_t shuffle, uint32_t n, uint32_t* out) {
  const rvt::PermKeys pk = rvt::perm_keys(seed, gene, shuffle);
  const int bits = rvt::perm_bits(n);
  for (uint32_t i = 0; i < n; ++i) out[i] = rvt::perm_index(i, n, bits, pk);
}

// ---- host_stage.h: the copy pool and the staging ring that feed the device from pageable memory -------------------------
// GB/s of `reps` copies of `bytes` bytes with a pool of `threads` threads (plain memory to plain memory: what the host
// side of a staged copy costs)
double hc_copy_rate(size_t bytes, int threads, int reps) {
  std::vector<char> src(bytes), dst(bytes);
  for (size_t i = 0; i < bytes; i += 4096) src[i] = (char)i;  // touch every page
  std::memset(dst.data(), 1, bytes);
  rvt::CopyPool pool(threads);
  pool.copy(dst.data(), src.data(), bytes);
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) pool.copy(dst.data(), src.data(), bytes);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return (double)bytes * reps / dt / 1e9;
}
// copy_gather with a fake device: n pieces of random lengths at increasing offsets with gaps; returns 0 when the "device"
// range holds every piece at its offset, zeros in the gaps and behind the last piece, and nothing beyond was touched.
int hc_stage_gather(int n, size_t max_piece, size_t chunk_bytes, int chunks, int threads, unsigned seed) {
  std::vector<size_t> len(n), off(n);
  size_t total = 0;
  unsigned x = seed * 2654435761u + 12345u;
  auto rnd = [&]() {
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
  };
  const size_t tail = 16;
  for (int i = 0; i < n; ++i) {
    len[i] = 1 + rnd() % max_piece;
    off[i] = total;
    total += (len[i] + 31) / 16 * 16;
  }
  std::vector<std::vector<unsigned char>> src(n);
  for (int i = 0; i < n; ++i) {
    src[i].resize(len[i]);
    for (auto& b : src[i]) b = (unsigned char)(1 + rnd() % 255);
  }
  std::vector<unsigned char> dev(total + 64, 0xEE);
  std::vector<std::vector<char>> mem((size_t)chunks, std::vector<char>(chunk_bytes, (char)0x77));
  rvt::StageRing ring;
  for (auto& m : mem) ring.chunk.push_back(m.data());
  ring.chunk_bytes = chunk_bytes;
  ring.wait = [](int) { return 0; };
  ring.send = [&](int k, size_t o, void* dst, size_t bytes) {
    std::memcpy(dst, ring.chunk[k] + o, bytes);
    return 0;
  };
  ring.send2d = [&](int, void*, size_t, size_t, size_t) { return 1; };
  ring.sent = [](int) { return 0; };
  std::vector<rvt::StageRing::Piece> pieces(n);
  for (int i = 0; i < n; ++i) pieces[i] = rvt::StageRing::Piece{off[i], src[i].data(), len[i]};
  rvt::CopyPool pool(threads);
  if (ring.copy_gather(dev.data(), pieces.data(), pieces.size(), tail, pool)) return 1;
  for (int i = 0; i < n; ++i) {
    if (std::memcmp(dev.data() + off[i], src[i].data(), len[i])) return 2;
    const size_t gap_end = (i + 1 < n) ? off[i + 1] : off[i] + len[i] + tail;
    for (size_t b = off[i] + len[i]; b < gap_end && b < off[i] + len[i] + tail; ++b)
      if (dev[b] != 0) return 3;  // (at least `tail` zero bytes behind every piece that has the room)
  }
  for (size_t b = off[n - 1] + len[n - 1] + tail; b < dev.size(); ++b)
    if (dev[b] != 0xEE) return 4;
  return 0;
}
// The staging ring with a fake device (plain memory): a 2-D copy of `rows` rows of `width` bytes (host pitch spitch,
// "device" pitch dpitch) through `chunks` chunks of `chunk_bytes`; returns 0 when every byte arrived where hipMemcpy2D
// would have put it and nothing else was touched.
int hc_stage_copy2d(size_t width, size_t rows, size_t spitch, size_t dpitch, size_t chunk_bytes, int chunks, int threads) {
  std::vector<unsigned char> src(spitch * rows), dev(dpitch * rows, 0xEE);
  for (size_t i = 0; i < src.size(); ++i) src[i] = (unsigned char)((i * 2654435761u) >> 13);
  std::vector<std::vector<char>> mem((size_t)chunks, std::vector<char>(chunk_bytes));
  rvt::StageRing ring;
  for (auto& m : mem) ring.chunk.push_back(m.data());
  ring.chunk_bytes = chunk_bytes;
  ring.wait = [](int) { return 0; };
  ring.send = [&](int k, size_t off, void* dst, size_t bytes) {
    std::memcpy(dst, ring.chunk[k] + off, bytes);
    return 0;
  };
  ring.send2d = [&](int k, void* dst, size_t dp, size_t w, size_t r) {
    for (size_t i = 0; i < r; ++i) std::memcpy((char*)dst + i * dp, ring.chunk[k] + i * w, w);
    return 0;
  };
  ring.sent = [](int) { return 0; };
  rvt::CopyPool pool(threads);
  if (ring.copy2d(dev.data(), dpitch, src.data(), spitch, width, rows, pool)) return 1;
  for (size_t r = 0; r < rows; ++r)
    for (size_t b = 0; b < dpitch; ++b) {
      const unsigned char want = b < width ? src[r * spitch + b] : 0xEE;
      if (dev[r * dpitch + b] != want) return 2;
    }
  return 0;
}

// ---- band_tiles.h: the tile lists of MetaCov's band products ---------------------------------------------------------------------
// the integer band (256 x 256 tiles): number of tiles; with out != null, out[3 t .. 3 t + 2] = row panel, column tile (in tiles of
// the window), index within the panel, in the order the kernels enumerate them
int hc_band_tiles(int H, int W, int halo, int* out) {
  const int n = rvt::band_tiles(H, W, halo);
  if (out) {
    int t = 0;
    for (int rp = 0; rp < (H + rvt::kBandBT - 1) / rvt::kBandBT; ++rp)
      for (int k = 0; k < rvt::band_panel_tiles(rp, W, halo); ++k, ++t) {
        out[3 * t] = rp;
        out[3 * t + 1] = rp + k;
        out[3 * t + 2] = k;
      }
  }
  return n;
}
int hc_band_tile_of(int h, int j, int W, int halo) { return rvt::band_tile_of(h, j, W, halo); }
long long hc_band_slices(int n_tiles, long long chunks, long long max_part_bytes) {
  return rvt::band_slices(n_tiles, chunks, (size_t)max_part_bytes);
}
// the fp64 product (256 x 128 tiles): number of tiles computed; out[2 t], out[2 t + 1] = row panel, column tile
int hc_gemm_f64_tiles(int M, int Ntot, int symmetric, int halo, int* out) {
  int nct = 0;
  const int n = rvt::gemm_f64_tiles(M, Ntot, symmetric != 0, &nct, halo);
  if (out) {
    int t = 0;
    const int nrp = (M + rvt::kGemmTileM - 1) / rvt::kGemmTileM;
    for (int rp = 0; rp < nrp; ++rp) {
      const int first = symmetric ? (rp * rvt::kGemmTileM) / rvt::kGemmTileN : 0;
      const int last = symmetric ? rvt::gemm_f64_panel_last(rp, nct, halo) : nct;
      for (int ct = first; ct < last; ++ct, ++t) {
        out[2 * t] = rp;
        out[2 * t + 1] = ct;
      }
    }
  }
  return n;
}

// pack_column_f64 with a chosen instruction set (0 scalar, 1 AVX2, 2 AVX-512; an ISA the CPU lacks falls back to scalar):
// returns ok; mu / has_mu / n_other as PackedColumn.  reps > 1: repeated (for a rate), the last result counts.
int hc_pack_column(const double* g, size_t n, unsigned char* out, size_t pitch, int isa, int reps, double* mu, int* has_mu,
                   long long* n_other) {
  rvt::PackedColumn r;
  for (int k = 0; k < (reps < 1 ? 1 : reps); ++k) r = rvt::pack_column_f64(g, n, out, pitch, nullptr, isa);
  *mu = r.mu;
  *has_mu = r.has_mu ? 1 : 0;
  *n_other = r.n_other;
  return r.ok ? 1 : 0;
}
// pack_column_i8 (int8 hard calls, negative = missing -> PLINK 2-bit codes) with a chosen instruction set (0 scalar, 1 AVX2)
int hc_pack_column_i8(const signed char* g, size_t n, unsigned char* out, size_t pitch, int isa, int reps) {
  bool ok = false;
  for (int k = 0; k < (reps < 1 ? 1 : reps); ++k) ok = rvt::pack_column_i8(g, n, out, pitch, isa);
  return ok ? 1 : 0;
}
int hc_cpu_has(int isa) {
#if defined(__x86_64__)
  if (isa == 1) return __builtin_cpu_supports("avx2") ? 1 : 0;
  if (isa == 2) return (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("bmi2")) ? 1 : 0;
#endif
  return isa == 0;
}
}  // extern "C"
