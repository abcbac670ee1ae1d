// tools/fdx_bench.hip — check and micro-benchmark of the float-digit dosage kernel (rvtests_amd/csrc/suffstat_fdx.hip.h) outside
// the engine.
//   fdx_bench check     small N (ragged ends, pad columns, flipped columns, values at the grid's limits, one off-grid value): every
//                       output against an exact evaluation on the host in 128-bit integers
//   fdx_bench [spread]  N = 500 000: algorithmic TB/s (8 N M + 8 N (d + 2) bytes per gene) per tile class; "spread": the widths of a
//                       512-gene batch with M ~ U{20..80}
// build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fdx_bench.hip -o tools/fdx_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../rvtests_amd/csrc/suffstat_fdx.hip.h"

using namespace rvt;
typedef __int128 i128;

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)

static inline unsigned long long mix(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ inline unsigned long long dmix(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// float-precision dosages of rare variants, as an 8-bit BGEN block gives them: p1 = float(v1) * s, p2 = float(v2) * s with
// s = float(1 / 255), dosage = p1 + 2 p2 in double (src/BGenGenotypeExtractor.cpp:413-478); most samples carry (255, 0, 0).
__global__ void fill_G(double* G, long long ld, long long N, int M, unsigned long long seed, double rate) {
  const long long total = ld * M;
  const float s = (float)(1.0 / 255.0);
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long i = idx % ld;
    const unsigned long long h = dmix(seed ^ (unsigned long long)idx * 0xD1B54A32D192ED03ull);
    const double a = (double)(h >> 40) * (1.0 / 16777216.0);
    unsigned v1 = 0, v2 = 0;
    if (a < rate) {                    // a carrier: probabilities blurred by the imputation
      v1 = 255u - (unsigned)((h >> 8) & 31u);
      v2 = (unsigned)((h >> 13) & 15u);
    } else if (a < 3 * rate) {         // a doubtful reference call
      v1 = (unsigned)((h >> 8) & 7u);
      v2 = (unsigned)((h >> 11) & 1u);
    }
    const float p1 = __fmul_rn((float)v1, s), p2 = __fmul_rn((float)v2, s);
    G[idx] = (i < N) ? (double)p1 + (double)p2 * 2.0 : 0.0;
  }
}

struct Null {
  long long N, ld;
  int d, ncols;
  std::vector<long long> X;          // [ncols][ld] integers (|x| < 2^38)
  std::vector<unsigned char> xq;     // device image
  double scale[16];
};
static Null make_null(long long N, int d, unsigned long long seed) {
  Null nl;
  nl.N = N;
  nl.ld = (N + 15) / 16 * 16;
  nl.d = d;
  nl.ncols = 2 * d + 3;  // [X_0 .. | res | 1 | low digits of X_0 .. res] as rvt_set_null lays them out
  const long long ld = nl.ld;
  nl.X.assign((size_t)nl.ncols * ld, 0);
  const long long ngroups = (ld + 31) / 32 + 8;
  nl.xq.assign((size_t)ngroups * kFdxPlanes * 4 * nl.ncols * 8, 0);
  for (long long i = 0; i < N; ++i)
    for (int k = 0; k < nl.ncols; ++k) {
      long long x = (k == d + 1) ? 1 : (k > d + 1) ? (long long)(mix(seed * 91 + i * 16 + k) % 256ull) - 128
                                   : (long long)(mix(seed * 77 + i * 16 + k) % (1ull << 39)) - (1ll << 38);
      nl.X[(size_t)k * ld + i] = x;
      const unsigned long long kb = (unsigned long long)(x + 0x8080808080ll);   // balanced base-256 digits: the biased bytes ^ 0x80
      const long long g = i >> 5, T = (i >> 4) & 1, q = (i >> 2) & 3, l = i & 3;
      for (int p = 0; p < kFdxPlanes; ++p)
        nl.xq[(((size_t)(g * kFdxPlanes + p) * 4 + q) * nl.ncols + k) * 8 + T * 4 + l] = (unsigned char)(((kb >> (8 * p)) & 0xff) ^ 0x80);
    }
  for (int k = 0; k < 16; ++k) nl.scale[k] = (k == d + 1) ? 1.0 : (k > d + 1) ? std::ldexp(1.0, -48 + (k - d - 2) % 3) : std::ldexp(1.0, -40 + k % 3);
  return nl;
}

struct DevGenes {
  std::vector<GeneDesc> gds;
  GeneDesc* dgd = nullptr;
  double *dG = nullptr, *parts = nullptr, *colstat = nullptr, *bparts = nullptr;
  unsigned* wflags = nullptr;
  size_t gstride = 0;
  int nw = 0, Mp = 0, Cp = 0, ngenes = 0;
  void free_all() { hipFree(dG); hipFree(parts); hipFree(colstat); hipFree(bparts); hipFree(wflags); hipFree(dgd); }
};
static DevGenes make_genes(long long N, long long ld, int d, int MT, int Mlo, int Mhi, int ngenes, int nw, long long spw, double rate) {
  DevGenes D;
  D.ngenes = ngenes;
  D.nw = nw;
  const int CTmax = (Mhi + d + 1 + 15) / 16;
  D.Mp = 16 * MT;
  D.Cp = 16 * CTmax;
  D.gstride = (size_t)ld * Mhi;
  CK(hipMalloc(&D.dG, sizeof(double) * D.gstride * ngenes));
  CK(hipMalloc(&D.parts, sizeof(double) * (size_t)ngenes * nw * D.Mp * D.Cp));
  CK(hipMalloc(&D.colstat, sizeof(double) * (size_t)ngenes * nw * kHcColstatRows * D.Mp));
  CK(hipMalloc(&D.wflags, sizeof(unsigned) * (size_t)ngenes * nw));
  CK(hipMalloc(&D.bparts, sizeof(double) * (size_t)ngenes * nw * 2 * (3 + d)));
  D.gds.resize(ngenes);
  for (int g = 0; g < ngenes; ++g) {
    GeneDesc& gd = D.gds[g];
    memset(&gd, 0, sizeof(gd));
    gd.G = D.dG + D.gstride * g;
    const int Mg = Mlo + (g * 7) % (Mhi - Mlo + 1);
    hipLaunchKernelGGL(fill_G, dim3(1024), dim3(256), 0, 0, D.dG + D.gstride * g, ld, N, Mg, 7ull + g, rate);
    gd.M = Mg; gd.MT = MT; gd.CT = (Mg + d + 1 + 15) / 16; gd.Mp = D.Mp; gd.Cp = 16 * gd.CT;
    gd.n_wparts = nw; gd.steps_per_wpart = (int)spw;
    gd.parts = D.parts + (size_t)g * nw * D.Mp * D.Cp;
    gd.colstat = D.colstat + (size_t)g * nw * kHcColstatRows * D.Mp;
    gd.wflags = D.wflags + (size_t)g * nw;
    gd.bparts = D.bparts + (size_t)g * nw * 2 * (3 + d);
    gd.n_bparts = nw; gd.hc = 2; gd.lat_den = 0x1p37;
    for (int j = 0; j < Mg; ++j)
      if (j % 11 == 3) gd.pflip[j >> 4] |= (unsigned short)(1u << (j & 15));
  }
  CK(hipMalloc(&D.dgd, sizeof(GeneDesc) * ngenes));
  CK(hipMemcpy(D.dgd, D.gds.data(), sizeof(GeneDesc) * ngenes, hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());
  return D;
}

typedef void (*fdx_kernel_t)(const GeneDesc*, NullTileF, long long, long long, int);
static fdx_kernel_t fdx_kernel(int MT) {
  switch (MT) {
    case 1: return gene_suffstat_fdx<1>;
    case 2: return gene_suffstat_fdx<2>;
    case 3: return gene_suffstat_fdx<3>;
    case 4: return gene_suffstat_fdx<4>;
    default: return gene_suffstat_fdx<5>;
  }
}
static double d128(i128 x) { return (double)x; }  // (one rounding)
// the kernel combines nine exact int32 order sums in fp64 from the top down: a few roundings of 2^-53 when the integer exceeds 2^53
static bool near(double got, double want) { return got == want || fabs(got - want) <= 1e-15 * fabs(want); }

static int check() {
  int bad_total = 0;
  const int d = 3;
  // (N, M, steps per wave-part, carrier rate, special: 1 = the grid's extreme values in column 0, 2 = one off-grid value)
  const struct { long long N; int M; int steps; double rate; int special; } cases[] = {
      {8192, 80, 512, 0.6, 0}, {5000, 80, 32, 0.05, 0}, {5000, 65, 48, 0.05, 1}, {3333, 50, 16, 0.05, 0}, {4097, 64, 32, 0.05, 0},
      {2600, 37, 64, 0.05, 0}, {3000, 20, 16, 0.05, 1}, {1000, 9, 8, 0.1, 0}, {777, 1, 8, 0.2, 0}, {6000, 33, 400, 0.05, 0}, {3000, 40, 24, 0.05, 2}};
  for (const auto& cs : cases) {
    const long long N = cs.N;
    const Null nl = make_null(N, d, 1234 + cs.M);
    const long long ld = nl.ld, nsteps = ld >> 4;
    const int M = cs.M, MT = (M + 15) / 16;
    const long long spw = cs.steps;
    const int nw = (int)((nsteps + spw - 1) / spw);
    DevGenes D = make_genes(N, ld, d, MT, M, M, 1, nw, spw, cs.rate);
    std::vector<double> G((size_t)ld * M);
    CK(hipMemcpy(G.data(), D.dG, sizeof(double) * G.size(), hipMemcpyDeviceToHost));
    if (cs.special == 1) {  // the limits of the grid: 2.0, 2^-37, 2 - 2^-37, the smallest 8-bit probability
      G[5] = 2.0; G[6] = 0x1p-37; G[7] = 2.0 - 0x1p-37; G[8] = (double)(float)(1.0 / 255.0); G[9] = 1.0; G[10] = 1.0 + 0x1p-37; G[11] = 1.0 - 0x1p-37;
      G[(size_t)3 * ld + 100] = 1.0; G[(size_t)3 * ld + 101] = 1.0 + 0x1p-37; G[(size_t)3 * ld + 102] = 1.0 - 0x1p-37;   // (column 3 is flipped)
    }
    if (cs.special == 2) G[(size_t)7 * ld + 1234] = 0.998;  // a decimal dosage: not on the grid
    CK(hipMemcpy(D.dG, G.data(), sizeof(double) * G.size(), hipMemcpyHostToDevice));
    unsigned char* dxq;
    CK(hipMalloc(&dxq, nl.xq.size()));
    CK(hipMemcpy(dxq, nl.xq.data(), nl.xq.size(), hipMemcpyHostToDevice));
    NullTileF nt;
    nt.xq = dxq;
    for (int k = 0; k < 16; ++k) nt.scale[k] = nl.scale[k];
    nt.ncols = nl.ncols;
    CK(hipMemset(D.parts, 0xff, sizeof(double) * (size_t)nw * D.Mp * D.Cp));
    hipLaunchKernelGGL(fdx_kernel(MT), dim3(nw, 1), dim3((kFdxNW + kFdxTW) * 64), 0, 0, D.dgd, nt, N, ld, d);
    CK(hipDeviceSynchronize());
    const GeneDesc& gd = D.gds[0];
    const int Mp = D.Mp, Cp = gd.Cp;
    std::vector<double> parts((size_t)nw * Mp * Cp), colstat((size_t)nw * kHcColstatRows * Mp), bparts((size_t)nw * 2 * (3 + d));
    std::vector<unsigned> wfl(nw);
    CK(hipMemcpy(parts.data(), D.parts, sizeof(double) * parts.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(colstat.data(), D.colstat, sizeof(double) * colstat.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(bparts.data(), D.bparts, sizeof(double) * bparts.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(wfl.data(), D.wflags, sizeof(unsigned) * nw, hipMemcpyDeviceToHost));
    unsigned fl = 0;
    for (int p = 0; p < nw; ++p) fl |= wfl[p];
    int bad = 0;
    auto expect = [&](bool ok, const char* what, int a, int b, double got, double want) {
      if (!ok && bad++ < 8) printf("   MISMATCH %s [%d,%d]: got %.17g want %.17g\n", what, a, b, got, want);
    };
    if (cs.special == 2) {
      expect(fl == 2u, "wflags (off-grid value)", 0, 0, fl, 2);
      printf("check N=%lld M=%d (MT %d): one off-grid value -> hand-back flag: %s\n", N, M, MT, bad ? "FAILED" : "ok");
      bad_total += bad;
      D.free_all();
      hipFree(dxq);
      continue;
    }
    // ---- host: exact integers K = g 2^37 ---------------------------------------------------------------------------
    std::vector<long long> K((size_t)N * M);
    for (int j = 0; j < M; ++j)
      for (long long i = 0; i < N; ++i) K[(size_t)i * M + j] = (long long)std::ldexp(G[(size_t)j * ld + i], 37);
    // every wave-part on its own: the kernel rounds once per part (a part's integer may exceed 2^53)
    auto part_range = [&](int p, long long& i0, long long& i1) {
      i0 = (long long)p * spw * 16;
      i1 = std::min<long long>(N, (long long)(p + 1) * spw * 16);
    };
    for (int p = 0; p < nw; ++p) {
      long long i0, i1;
      part_range(p, i0, i1);
      for (int j = 0; j < M; ++j) {
        for (int k = (j >> 4) << 4; k < M; ++k) {
          i128 s = 0;
          for (long long i = i0; i < i1; ++i) s += (i128)K[(size_t)i * M + j] * K[(size_t)i * M + k];
          const double got = parts[((size_t)p * Mp + j) * Cp + k];
          expect(near(got, d128(s)), "K'K", j, k, got, d128(s));
        }
        for (int k = 0; k < 16 && M + k < Cp; ++k) {
          i128 s = 0, sl = 0;
          if (k <= d)
            for (long long i = i0; i < i1; ++i) {
              s += (i128)K[(size_t)i * M + j] * nl.X[(size_t)k * ld + i];
              sl += (i128)K[(size_t)i * M + j] * nl.X[(size_t)(k + d + 2) * ld + i];
            }
          const double got = parts[((size_t)p * Mp + j) * Cp + M + k];
          const double want = (k <= d) ? d128(s) * (nl.scale[k] * 0x1p-37) + d128(sl) * (nl.scale[k + d + 2] * 0x1p-37) : 0.0;
          expect(near(got, want), "K'[X|res]", j, k, got, want);
        }
        long long s = 0;
        double mn = INFINITY, mx = -INFINITY;
        for (long long i = i0; i < i1; ++i) {
          s += K[(size_t)i * M + j];
          mn = fmin(mn, G[(size_t)j * ld + i]);
          mx = fmax(mx, G[(size_t)j * ld + i]);
        }
        const double* c = colstat.data() + (size_t)p * kHcColstatRows * Mp;
        expect(c[j] == (double)s, "colsum", j, p, c[j], (double)s);
        expect(c[Mp + j] == mn && c[2 * Mp + j] == mx, "min/max", j, p, c[Mp + j], mn);
      }
      // burden sums
      long long U[2] = {0, 0}, cc2[2] = {0, 0}, cnt = 0;
      i128 cx[2][16];
      for (int t = 0; t < 2; ++t)
        for (int k = 0; k < 16; ++k) cx[t][k] = 0;
      for (long long i = i0; i < i1; ++i) {
        int n = 0;
        for (int j = 0; j < M; ++j) {
          const bool flip = (gd.pflip[j >> 4] >> (j & 15)) & 1;
          const double g = G[(size_t)j * ld + i];
          n += flip ? ((int)(2.0 - g) > 0) : ((int)g > 0);
        }
        const long long c[2] = {n > 0 ? 1 : 0, n};
        cnt += n > 0;
        for (int t = 0; t < 2; ++t) {
          cc2[t] += c[t] * c[t];
          for (int k = 0; k <= d; ++k) {
            cx[t][k] += (i128)c[t] * nl.X[(size_t)k * ld + i];
            cx[t][8 + k] += (i128)c[t] * nl.X[(size_t)(k + d + 2) * ld + i];
          }
        }
      }
      (void)U;
      const int rl = 3 + d;
      for (int t = 0; t < 2; ++t) {
        const double* b = bparts.data() + ((size_t)p * 2 + t) * rl;
        auto both = [&](int k) { return d128(cx[t][k]) * nl.scale[k] + d128(cx[t][8 + k]) * nl.scale[k + d + 2]; };
        expect(near(b[0], both(d)), "burden U", t, p, b[0], both(d));
        expect(b[1] == (double)cc2[t], "burden c'c", t, p, b[1], (double)cc2[t]);
        expect(b[2] == (double)cnt, "burden count", t, p, b[2], (double)cnt);
        for (int k = 0; k < d; ++k) expect(near(b[3 + k], both(k)), "burden c'X", t, k, b[3 + k], both(k));
      }
    }
    expect(fl == 0u, "wflags", 0, 0, fl, 0);
    printf("check N=%lld M=%d (MT %d) parts=%d x %lld steps%s: %s\n", N, M, MT, nw, spw, cs.special == 1 ? ", grid limits" : "", bad ? "FAILED" : "ok");
    bad_total += bad;
    D.free_all();
    hipFree(dxq);
  }
  return bad_total ? 1 : 0;
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  bool spread = false;
  for (int a = 1; a < argc; ++a) {
    if (!strcmp(argv[a], "check")) return check();
    if (!strcmp(argv[a], "spread")) spread = true;
  }
  const long long N = 500000;
  const int d = 3;
  const Null nl = make_null(N, d, 99);
  const long long ld = nl.ld, nsteps = ld >> 4;
  unsigned char* dxq;
  CK(hipMalloc(&dxq, nl.xq.size()));
  CK(hipMemcpy(dxq, nl.xq.data(), nl.xq.size(), hipMemcpyHostToDevice));
  NullTileF nt;
  nt.xq = dxq;
  for (int k = 0; k < 16; ++k) nt.scale[k] = nl.scale[k];
  nt.ncols = nl.ncols;
  const int Ms[] = {12, 28, 44, 60, 76};
  for (int MT = 1; MT <= 5; ++MT) {
    if (spread && MT == 1) continue;
    const int Mlo = spread ? (MT == 2 ? 20 : 16 * (MT - 1) + 1) : Ms[MT - 1], Mhi = spread ? 16 * MT : Ms[MT - 1];
    const int ngenes = 48;
    const long long spw = 512;
    const int nw = (int)((nsteps + spw - 1) / spw);
    DevGenes D = make_genes(N, ld, d, MT, Mlo, Mhi, ngenes, nw, spw, 0.02);
    double bytes = 0;
    for (const GeneDesc& g : D.gds) bytes += 8.0 * N * g.M + 8.0 * N * (d + 2);
#ifdef FDX_PROF
    unsigned long long* dprof;
    CK(hipMalloc(&dprof, 32 * 8));
    CK(hipMemset(dprof, 0, 32 * 8));
    for (GeneDesc& g : D.gds) g.dbg_cmc = reinterpret_cast<double*>(dprof);
    CK(hipMemcpy(D.dgd, D.gds.data(), sizeof(GeneDesc) * ngenes, hipMemcpyHostToDevice));
#endif
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int k = 0; k < 3; ++k)
        hipLaunchKernelGGL(fdx_kernel(MT), dim3(nw, ngenes), dim3((kFdxNW + kFdxTW) * 64), 0, 0, D.dgd, nt, N, ld, d);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    printf("bench M=%d..%d MT=%d fdx parts=%d: %.3f ms per %d genes, %.2f TB/s algorithmic\n", Mlo, Mhi, MT, nw, ms, ngenes, bytes / (ms * 1e-3) / 1e12);
#ifdef FDX_PROF
    {
      unsigned long long hp[32];
      CK(hipMemcpy(hp, dprof, sizeof(hp), hipMemcpyDeviceToHost));
      const double niter = (double)nw * ngenes * 6 * (double)(spw / kFdxIterSteps);
      for (int w = 0; w < 4; ++w)
        printf("   loader %d cycles per iteration: work %.0f  vmcnt wait %.0f  barrier %.0f\n", w, hp[w * 4] / niter, hp[w * 4 + 2] / niter, hp[w * 4 + 1] / niter);
      for (int w = 4; w < 4 + kFdxTW; ++w)
        printf("   tile wave %d cycles per iteration: tiles %.0f  barrier %.0f\n", w - 4, hp[w * 4] / niter, hp[w * 4 + 1] / niter);
    }
#endif
    D.free_all();
  }
  return 0;
}
