// tools/k2hc_bench.hip — standalone check + micro-benchmark of the hard-call sufficient-statistics kernel
// (rvtests_amd/csrc/suffstat_hc.hip.h) outside the engine.
//   check:  small N, every tile class, odd N, a flipped column: partial results summed on the host and compared with a
//           plain host loop (integers exact, fp64 sums to 1e-12)
//   bench:  N = 500 000, per tile class, nt on/off: algorithmic TB/s (8 N M + 8 N (d + 2) bytes per gene)
// build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/k2hc_bench.hip -o tools/k2hc_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "../rvtests_amd/csrc/suffstat_hc.hip.h"

using namespace rvt;

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)

__device__ __host__ inline unsigned long long mix(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// G[j][i] ~ Binomial(2, maf_j), pad rows zero; maf_j from a hash (log-uniform 5e-4..5e-2); column `flipcol` gets maf 0.9
__global__ void fill_G(double* G, long long N, long long ld, int M, unsigned long long seed, int flipcol,
                       double miss = 0.0, long long gene_ld = 0) {
  const long long total = ld * M;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long j = idx / ld, i = idx % ld;
    double g = 0.0;
    if (i < N) {
      const double u = (double)(mix(seed * 1315423911ull + j) >> 11) * (1.0 / 9007199254740992.0);
      double maf = pow(10.0, -3.3 + 2.0 * u);
      if (j == flipcol) maf = 0.9;
      const unsigned long long h = mix(seed ^ (unsigned long long)(j * 1000003ll + 7) * 0x9E3779B97F4A7C15ull ^ (unsigned long long)i * 0xD1B54A32D192ED03ull);
      const double a = (double)(h >> 40) * (1.0 / 16777216.0), b = (double)((h >> 16) & 0xffffff) * (1.0 / 16777216.0);
      g = (a < maf ? 1.0 : 0.0) + (b < maf ? 1.0 : 0.0);
      if (miss > 0.0) {  // a "mean-imputed" entry: one value per column (per gene when genes sit back to back)
        const long long gi = gene_ld > 0 ? i / gene_ld : 0;
        const unsigned long long h2 = mix(h ^ 0xA5A5A5A5DEADBEEFull);
        if ((double)(h2 >> 11) * (1.0 / 9007199254740992.0) < miss)
          g = 2.0 * maf + 1e-3 * (double)((j * 7 + gi * 13) % 97) + 1.0 / 3.0 * 1e-5;
      }
    }
    G[idx] = g;
  }
}

__global__ void fill_null(double* T, long long N, long long ld, int d, unsigned long long seed) {
  const long long total = ld * (d + 2);
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long k = idx / ld, i = idx % ld;
    double x = 0.0;
    if (i < N && k <= d) {
      const unsigned long long h = mix(seed + 77 * k + (unsigned long long)i * 0x2545F4914F6CDD1Dull);
      x = (k == 0) ? 1.0 : ((double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 3.0;
    }
    T[idx] = x;
  }
}


// candidate configurations per tile class: (MT, DEPTH, WAVES)
typedef void (*hc_kernel_t)(const GeneDesc*, NullTile, long long, long long, int);
struct HcCfg {
  int MT, depth, waves;
  hc_kernel_t k[2];  // nt off / on
};
#define CFG(mt, dp, w) {mt, dp, w, {gene_suffstat_hc<mt, dp, w, false>, gene_suffstat_hc<mt, dp, w, true>}}
static const HcCfg kCfgs[] = {
    CFG(1, 2, 4), CFG(1, 2, 5), CFG(2, 2, 3), CFG(2, 3, 2), CFG(3, 2, 2), CFG(3, 3, 1),
    CFG(4, 2, 2), CFG(4, 3, 1), CFG(5, 2, 1), CFG(5, 1, 1), CFG(5, 1, 2), CFG(6, 2, 1), CFG(6, 1, 1),
};
static void launch_cfg(const HcCfg& c, int nt_on, dim3 grid, const GeneDesc* dgd, NullTile nt, long long N, long long ld, int d) {
  hipLaunchKernelGGL(c.k[nt_on], grid, dim3(64), 0, 0, dgd, nt, N, ld, d);
}

struct Gene {
  int M, MT, CT, Mp, Cp;
  double* dG;
  double *parts, *colstat, *bparts;
  std::vector<int> pflip;
};

static void choose(long long ld, int* n_wparts, int* spw) {
  const long long nsteps = ld >> 4;
  long long s = (nsteps + 127) / 128;
  if (s < 64) s = 64;
  s = (s + kHcStepUnit - 1) / kHcStepUnit * kHcStepUnit;
  *spw = (int)s;
  *n_wparts = (int)((nsteps + s - 1) / s);
}

int main(int argc, char** argv) {
  const bool bench_only = argc > 1 && !strcmp(argv[1], "bench");
  const bool check_only = argc > 1 && !strcmp(argv[1], "check");
  CK(hipSetDevice(0));
  int fails = 0;
  // ------------------------------------------------------------------------------------------------ check
  if (!bench_only) {
    const int d = 3;
    struct Case { long long N; int M; int flip; double miss; int poison; };  // poison 1: a second non-hard value in column 0, 2: -inf
    const Case cases[] = {{4000, 50, 3, 0.0, 0},  {4099, 1, -1, 0.0, 0},   {777, 16, 0, 0.0, 0},    {5003, 17, -1, 0.0, 0},
                          {3000, 33, 20, 0.0, 0}, {9001, 64, 63, 0.0, 0},  {2049, 80, 5, 0.0, 0},   {6000, 96, 95, 0.0, 0},
                          {640, 7, -1, 0.0, 0},   {50000, 30, -1, 0.0, 0}, {4000, 50, 3, 0.01, 0},  {4099, 1, -1, 0.05, 0},
                          {5003, 17, 2, 0.002, 0}, {3000, 33, 20, 0.3, 0}, {9001, 64, 63, 0.001, 0}, {2049, 80, 5, 0.01, 0},
                          {6000, 96, 95, 0.004, 0}, {50000, 30, -1, 0.001, 0}, {4000, 50, 3, 0.01, 1}, {4000, 20, -1, 0.0, 2},
                          {70000, 40, -1, 1.0, 0}};
    for (const Case& cs : cases) {
      const long long N = cs.N, ld = (N + 15) / 16 * 16;
      const int M = cs.M, MT = (M + 15) / 16, CT = (M + d + 1 + 15) / 16, Mp = 16 * MT, Cp = 16 * CT;
      int nw, spw;
      choose(ld, &nw, &spw);
      const int pqw = hc_pq_words(MT);
      double *dG, *dT, *parts, *colstat, *bparts;
      unsigned *pq, *wflags;
      CK(hipMalloc(&dG, sizeof(double) * ld * M));
      CK(hipMalloc(&dT, sizeof(double) * ld * (d + 2)));
      CK(hipMalloc(&parts, sizeof(double) * (size_t)nw * Mp * Cp));
      CK(hipMalloc(&colstat, sizeof(double) * (size_t)nw * kHcColstatRows * Mp));
      CK(hipMalloc(&bparts, sizeof(double) * (size_t)nw * 2 * (3 + d)));
      CK(hipMalloc(&pq, sizeof(unsigned) * (size_t)nw * pqw));
      CK(hipMalloc(&wflags, sizeof(unsigned) * (size_t)nw));
      hipLaunchKernelGGL(fill_G, dim3(1024), dim3(256), 0, 0, dG, N, ld, M, 1234ull + M, cs.flip, cs.miss, 0ll);
      hipLaunchKernelGGL(fill_null, dim3(256), dim3(256), 0, 0, dT, N, ld, d, 99ull);
      std::vector<double> G((size_t)ld * M), T((size_t)ld * (d + 2));
      CK(hipMemcpy(G.data(), dG, sizeof(double) * ld * M, hipMemcpyDeviceToHost));
      if (cs.poison == 1) G[(size_t)0 * ld + N / 2] = 0.77, G[(size_t)0 * ld + N / 3] = 0.78;
      if (cs.poison == 2) G[(size_t)(M - 1) * ld + N / 2] = -INFINITY;
      if (cs.poison) CK(hipMemcpy(dG, G.data(), sizeof(double) * ld * M, hipMemcpyHostToDevice));
      CK(hipMemcpy(T.data(), dT, sizeof(double) * ld * (d + 2), hipMemcpyDeviceToHost));
      GeneDesc gd;
      memset(&gd, 0, sizeof(gd));
      gd.G = dG;
      gd.M = M; gd.MT = MT; gd.CT = CT; gd.Mp = Mp; gd.Cp = Cp;
      gd.n_wparts = nw; gd.steps_per_wpart = spw;
      gd.parts = parts; gd.colstat = colstat; gd.bparts = bparts;
      gd.pq = pq; gd.wflags = wflags;
      gd.n_bparts = nw; gd.hc = 1;
      auto hard = [](double g) { return g == 0.0 || g == 1.0 || g == 2.0; };
      std::vector<int> flip(M, 0);
      for (int j = 0; j < M; ++j) {
        double s = 0;
        for (long long i = 0; i < N; ++i) s += G[(size_t)j * ld + i];
        flip[j] = s > (double)N;   // the host predicts from the allele frequency; here exactly
        if (flip[j]) gd.pflip[j >> 4] |= (unsigned short)(1u << (j & 15));
      }
      GeneDesc* dgd;
      CK(hipMalloc(&dgd, sizeof(gd)));
      CK(hipMemcpy(dgd, &gd, sizeof(gd), hipMemcpyHostToDevice));
      NullTile nt{dT, d + 2};
      // host reference of the integer pieces
      std::vector<long long> HH((size_t)M * M, 0), PP((size_t)M * M, 0), QQ((size_t)M * M, 0);
      std::vector<double> mu(M, 0.0);
      std::vector<long long> cm(M, 0);
      {
        std::vector<unsigned char> Hc((size_t)N * M), Mc((size_t)N * M);
        for (int j = 0; j < M; ++j)
          for (long long i = 0; i < N; ++i) {
            const double g = G[(size_t)j * ld + i];
            const bool hd = hard(g);
            Hc[(size_t)i * M + j] = hd ? (unsigned char)g : 0;
            Mc[(size_t)i * M + j] = hd ? 0 : 1;
            if (!hd) { mu[j] = g; cm[j]++; }
          }
        for (long long i = 0; i < N; ++i) {
          const unsigned char* hr = &Hc[(size_t)i * M];
          const unsigned char* mr = &Mc[(size_t)i * M];
          for (int a = 0; a < M; ++a) {
            if (!hr[a] && !mr[a]) continue;
            for (int b2 = 0; b2 < M; ++b2) {
              HH[(size_t)a * M + b2] += hr[a] * hr[b2];
              PP[(size_t)a * M + b2] += hr[a] * mr[b2];
              QQ[(size_t)a * M + b2] += mr[a] * mr[b2];
            }
          }
        }
      }
      for (const HcCfg& cf : kCfgs) {
      if (cf.MT != MT) continue;
      CK(hipMemset(parts, 0xff, sizeof(double) * (size_t)nw * Mp * Cp));
      CK(hipMemset(colstat, 0xff, sizeof(double) * (size_t)nw * kHcColstatRows * Mp));
      CK(hipMemset(bparts, 0xff, sizeof(double) * (size_t)nw * 2 * (3 + d)));
      CK(hipMemset(pq, 0xff, sizeof(unsigned) * (size_t)nw * pqw));
      CK(hipMemset(wflags, 0xff, sizeof(unsigned) * (size_t)nw));
      launch_cfg(cf, 0, dim3(nw, 1), dgd, nt, N, ld, d);
      CK(hipDeviceSynchronize());
      std::vector<double> hp((size_t)nw * Mp * Cp), hc((size_t)nw * kHcColstatRows * Mp), hb((size_t)nw * 2 * (3 + d));
      std::vector<unsigned> hq((size_t)nw * pqw), hw(nw);
      CK(hipMemcpy(hp.data(), parts, sizeof(double) * hp.size(), hipMemcpyDeviceToHost));
      CK(hipMemcpy(hc.data(), colstat, sizeof(double) * hc.size(), hipMemcpyDeviceToHost));
      CK(hipMemcpy(hb.data(), bparts, sizeof(double) * hb.size(), hipMemcpyDeviceToHost));
      CK(hipMemcpy(hq.data(), pq, sizeof(unsigned) * hq.size(), hipMemcpyDeviceToHost));
      CK(hipMemcpy(hw.data(), wflags, sizeof(unsigned) * hw.size(), hipMemcpyDeviceToHost));
      // reduce
      std::vector<double> R((size_t)Mp * Cp, 0.0);
      for (int p = 0; p < nw; ++p)
        for (int i = 0; i < Mp; ++i)
          for (int j = 0; j < Cp; ++j)
            if ((j >> 4) >= (i >> 4) && (j < M + d + 1)) R[(size_t)i * Cp + j] += hp[((size_t)p * Mp + i) * Cp + j];
      unsigned flag_or = 0;
      for (int p = 0; p < nw; ++p) flag_or |= hw[p];
      // masked tiles: element (row, col) of tile t sits in word (t * 2 + (reg >> 1)) * 64 + lane, half reg & 1,
      // lane = 16 * (row_in_tile >> 2) + col_in_tile, reg = row_in_tile & 3
      auto pq_get = [&](int p, int tile, int ri, int ci) -> long long {
        const int lane = 16 * (ri >> 2) + ci, reg = ri & 3;
        const unsigned w = hq[(size_t)p * pqw + (tile * 2 + (reg >> 1)) * 64 + lane];
        return (long long)((w >> (16 * (reg & 1))) & 0xffffu);
      };
      std::vector<long long> Pp((size_t)Mp * Mp, 0), Qd((size_t)Mp * Mp, 0);
      for (int p = 0; p < nw; ++p) {
        if (!(hw[p] & 1u)) continue;
        for (int r = 0; r < MT; ++r)
          for (int c = 0; c < MT; ++c)
            for (int ri = 0; ri < 16; ++ri)
              for (int ci = 0; ci < 16; ++ci) Pp[(size_t)(r * 16 + ri) * Mp + c * 16 + ci] += pq_get(p, r * MT + c, ri, ci);
        int t = MT * MT;
        for (int r = 0; r < MT; ++r)
          for (int c = r; c < MT; ++c, ++t)
            for (int ri = 0; ri < 16; ++ri)
              for (int ci = 0; ci < 16; ++ci) {
                const long long qv = pq_get(p, t, ri, ci);
                Qd[(size_t)(r * 16 + ri) * Mp + c * 16 + ci] += qv;
                if (c != r) Qd[(size_t)(c * 16 + ci) * Mp + r * 16 + ri] += qv;
              }
      }
      // column statistics
      std::vector<double> muk(M, 0.0);
      std::vector<long long> cmk(M, 0);
      int badstat = 0, inconsistent = 0;
      for (int j = 0; j < M; ++j) {
        double s = 0, mn = INFINITY, mx = -INFINITY, s0 = 0, mn0 = INFINITY, mx0 = -INFINITY;
        unsigned long long orb = 0, andb = ~0ull;
        for (int p = 0; p < nw; ++p) {
          const double* c0 = &hc[(size_t)p * kHcColstatRows * Mp];
          s += c0[j];
          mn = fmin(mn, c0[Mp + j]);
          mx = fmax(mx, c0[2 * Mp + j]);
          cmk[j] += (long long)c0[3 * Mp + j];
          unsigned long long b1, b2;
          memcpy(&b1, &c0[4 * Mp + j], 8);
          memcpy(&b2, &c0[5 * Mp + j], 8);
          orb |= b1;
          andb &= b2;
        }
        if (cmk[j] > 0) {
          if (orb != andb) ++inconsistent;
          memcpy(&muk[j], &orb, 8);
          s += (double)cmk[j] * muk[j];
          mn = fmin(mn, muk[j]);
          mx = fmax(mx, muk[j]);
        }
        for (long long i = 0; i < N; ++i) {
          const double g = G[(size_t)j * ld + i];
          s0 += g;
          mn0 = fmin(mn0, g);
          mx0 = fmax(mx0, g);
        }
        if (cs.poison) continue;
        if (fabs(s - s0) > 1e-9 * fmax(1.0, fabs(s0)) || mn != mn0 || mx != mx0 || cmk[j] != cm[j] || (cm[j] > 0 && muk[j] != mu[j])) ++badstat;
      }
      if (cs.poison) {
        const bool ok = (cs.poison == 1) ? (inconsistent == 1) : ((flag_or & 2u) != 0);
        printf("check N=%lld M=%d (MT=%d depth %d waves %d) poison %d: inconsistent columns %d, flags %u  %s\n", N, M, MT,
               cf.depth, cf.waves, cs.poison, inconsistent, flag_or, ok ? "OK" : "FAIL");
        if (!ok) ++fails;
        continue;
      }
      double worstS = 0, worstT = 0;
      long long worstI = 0;
      for (int a = 0; a < M; ++a) {
        for (int b = a; b < M; ++b) {
          if ((b >> 4) < (a >> 4)) continue;
          const long long q = Qd[(size_t)a * Mp + b];
          const long long pab = Pp[(size_t)a * Mp + b] - 4 * q, pba = Pp[(size_t)b * Mp + a] - 4 * q;
          const long long hh = (long long)R[(size_t)a * Cp + b] - 4 * (pab + pba) - 16 * q;
          worstI = std::max(worstI, std::llabs(hh - HH[(size_t)a * M + b]));
          worstI = std::max(worstI, std::llabs(pab - PP[(size_t)a * M + b]));
          worstI = std::max(worstI, std::llabs(pba - PP[(size_t)b * M + a]));
          worstI = std::max(worstI, std::llabs(q - QQ[(size_t)a * M + b]));
          const double sk = (double)hh + muk[b] * (double)pab + muk[a] * (double)pba + muk[a] * muk[b] * (double)q;
          long double s = 0;
          for (long long i = 0; i < N; ++i) s += (long double)G[(size_t)a * ld + i] * G[(size_t)b * ld + i];
          worstS = fmax(worstS, fabs((double)(s - sk)) / fmax(1.0, fabs((double)s)));
        }
        for (int k = 0; k <= d; ++k) {
          double s = 0, sa = 0;
          for (long long i = 0; i < N; ++i) {
            s += G[(size_t)a * ld + i] * T[(size_t)k * ld + i];
            sa += fabs(G[(size_t)a * ld + i] * T[(size_t)k * ld + i]);
          }
          worstT = fmax(worstT, fabs(s - R[(size_t)a * Cp + M + k]) / fmax(sa, 1e-300));
        }
      }
      // burden reference
      const int rl = 3 + d;
      std::vector<double> br(2 * rl, 0.0), bg(2 * rl, 0.0);
      std::vector<int> poly(M, 0);
      for (int j = 0; j < M; ++j) {
        double mn = INFINITY, mx = -INFINITY;
        for (long long i = 0; i < N; ++i) {
          mn = fmin(mn, G[(size_t)j * ld + i]);
          mx = fmax(mx, G[(size_t)j * ld + i]);
        }
        poly[j] = mn != mx;
      }
      for (long long i = 0; i < N; ++i) {
        int n = 0;
        for (int j = 0; j < M; ++j) {
          const double g = G[(size_t)j * ld + i], gf = flip[j] ? 2.0 - g : g;
          if (poly[j] && (int)gf > 0) ++n;
        }
        const double c[2] = {n > 0 ? 1.0 : 0.0, (double)n};
        for (int t = 0; t < 2; ++t) {
          br[t * rl + 0] += c[t] * T[(size_t)d * ld + i];
          br[t * rl + 1] += c[t] * c[t];
          br[t * rl + 2] += (c[t] != 0.0);
          for (int k = 0; k < d; ++k) br[t * rl + 3 + k] += c[t] * T[(size_t)k * ld + i];
        }
      }
      for (int p = 0; p < nw; ++p)
        for (int k = 0; k < 2 * rl; ++k) bg[k] += hb[(size_t)p * 2 * rl + k];
      double worstB = 0;
      for (int k = 0; k < 2 * rl; ++k) worstB = fmax(worstB, fabs(bg[k] - br[k]) / fmax(fabs(br[k]), 1.0));
      const bool ok = worstI == 0 && worstS < 1e-14 && worstT < 1e-12 && badstat == 0 && worstB < 1e-11 &&
                      inconsistent == 0 && (flag_or & 2u) == 0 && ((flag_or & 1u) != 0) == (cs.miss > 0);
      printf("check N=%lld M=%d miss %.3g (MT=%d depth %d waves %d, wparts=%d x %d steps) flipcol=%d: int pieces abs %lld  S rel %.3g  T rel %.3g  colstat bad %d  burden rel %.3g  flags %u  %s\n",
             N, M, cs.miss, MT, cf.depth, cf.waves, nw, spw, cs.flip, worstI, worstS, worstT, badstat, worstB, flag_or,
             ok ? "OK" : "FAIL");
      if (!ok) ++fails;
      }
      CK(hipFree(dG)); CK(hipFree(dT)); CK(hipFree(parts)); CK(hipFree(colstat)); CK(hipFree(bparts));
      CK(hipFree(pq)); CK(hipFree(wflags)); CK(hipFree(dgd));
    }
  }
  if (check_only) return fails ? 1 : 0;
  // ------------------------------------------------------------------------------------------------ bench
  {
    const long long N = 500000, ld = (N + 15) / 16 * 16;
    const int d = 3;
    int nw, spw;
    choose(ld, &nw, &spw);
    double* dT;
    CK(hipMalloc(&dT, sizeof(double) * ld * (d + 2)));
    hipLaunchKernelGGL(fill_null, dim3(1024), dim3(256), 0, 0, dT, N, ld, d, 99ull);
    NullTile nt{dT, d + 2};
    const int Ms[] = {12, 28, 44, 50, 60, 76, 92};
    const int ngenes = 64;
    const double misses[] = {0.0, 0.001, 0.01};
    for (int M : Ms) {
      const int MT = (M + 15) / 16, CT = (M + d + 1 + 15) / 16, Mp = 16 * MT, Cp = 16 * CT;
      const int pqw = hc_pq_words(MT);
      double* dG;  // one allocation, genes back to back
      const size_t gstride = (size_t)ld * M;
      CK(hipMalloc(&dG, sizeof(double) * gstride * ngenes));
      double *parts, *colstat, *bparts;
      unsigned *pq, *wflags;
      CK(hipMalloc(&parts, sizeof(double) * (size_t)ngenes * nw * Mp * Cp));
      CK(hipMalloc(&colstat, sizeof(double) * (size_t)ngenes * nw * kHcColstatRows * Mp));
      CK(hipMalloc(&bparts, sizeof(double) * (size_t)ngenes * nw * 2 * (3 + d)));
      CK(hipMalloc(&pq, sizeof(unsigned) * (size_t)ngenes * nw * pqw));
      CK(hipMalloc(&wflags, sizeof(unsigned) * (size_t)ngenes * nw));
      std::vector<GeneDesc> gds(ngenes);
      for (int g = 0; g < ngenes; ++g) {
        GeneDesc& gd = gds[g];
        memset(&gd, 0, sizeof(gd));
        gd.G = dG + gstride * g;
        gd.M = M; gd.MT = MT; gd.CT = CT; gd.Mp = Mp; gd.Cp = Cp;
        gd.n_wparts = nw; gd.steps_per_wpart = spw;
        gd.parts = parts + (size_t)g * nw * Mp * Cp;
        gd.colstat = colstat + (size_t)g * nw * kHcColstatRows * Mp;
        gd.bparts = bparts + (size_t)g * nw * 2 * (3 + d);
        gd.pq = pq + (size_t)g * nw * pqw;
        gd.wflags = wflags + (size_t)g * nw;
        gd.n_bparts = nw; gd.hc = 1;
      }
      GeneDesc* dgd;
      CK(hipMalloc(&dgd, sizeof(GeneDesc) * ngenes));
      CK(hipMemcpy(dgd, gds.data(), sizeof(GeneDesc) * ngenes, hipMemcpyHostToDevice));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (double miss : misses) {
        // (genes back to back in one allocation of M columns x ngenes * ld rows: any 0/1/2 data of the right size)
        hipLaunchKernelGGL(fill_G, dim3(4096), dim3(256), 0, 0, dG, ld * (long long)ngenes, ld * (long long)ngenes, M, 7ull, -1,
                           miss, ld * (long long)ngenes);
        CK(hipDeviceSynchronize());
        for (const HcCfg& cf : kCfgs) {
          if (cf.MT != MT) continue;
          auto launch = [&]() { launch_cfg(cf, 0, dim3(nw, ngenes), dgd, nt, N, ld, d); };
          launch();
          CK(hipDeviceSynchronize());
          const int reps = 5;
          CK(hipEventRecord(e0, 0));
          for (int r = 0; r < reps; ++r) launch();
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          const double bytes = (8.0 * N * M + 8.0 * N * (d + 2)) * ngenes * reps;
          printf("bench M=%d MT=%d depth=%d waves=%d miss=%.3g: %.3f ms per %d genes, %.2f TB/s algorithmic\n", M, MT, cf.depth,
                 cf.waves, miss, ms / reps, ngenes, bytes / (ms * 1e-3) / 1e12);
        }
      }
      CK(hipFree(dG)); CK(hipFree(parts)); CK(hipFree(colstat)); CK(hipFree(bparts)); CK(hipFree(dgd));
      CK(hipFree(pq)); CK(hipFree(wflags));
    }
  }
  return fails ? 1 : 0;
}
