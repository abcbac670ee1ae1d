// tools/hcx_bench.hip — check and micro-benchmark of the workgroup-cooperative weighted hard-call kernel
// (rvtests_amd/csrc/suffstat_hcx.hip.h) outside the engine.
//   hcx_bench check            small N (ragged end, masked entries, pad columns): every output against an exact integer
//                              evaluation on the host (the kernel's arithmetic is integer: equality, not tolerance)
//   hcx_bench [spread] [miss]  N = 200 000: algorithmic TB/s (8 N M + 8 N (d + 4) bytes per gene) per tile class beside the
//                              one-wave kernel gene_suffstat_hcw; "spread": the widths of a 512-gene batch with M ~ U{20..80};
//                              "miss": 0.1 % of the entries of every gene are mean-imputed
// build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hcx_bench.hip -o tools/hcx_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../rvtests_amd/csrc/suffstat_hcx.hip.h"

using namespace rvt;

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
// hard calls with rare alleles; miss_per_million of the entries replaced by the column's "mean" mu_j = 0.01 (j % 7 + 1)
__global__ void fill_G(double* G, long long ld, long long N, int M, unsigned long long seed, int miss_per_million, double rate) {
  const long long total = ld * M;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long i = idx % ld;
    const int j = (int)(idx / ld);
    const unsigned long long h = dmix(seed ^ (unsigned long long)idx * 0xD1B54A32D192ED03ull);
    const double a = (double)(h >> 40) * (1.0 / 16777216.0), b = (double)((h >> 16) & 0xffffff) * (1.0 / 16777216.0);
    double g = (a < rate ? 1.0 : 0.0) + (b < rate ? 1.0 : 0.0);
    if ((int)(dmix(h) % 1000000ull) < miss_per_million) g = 0.01 * (j % 7 + 1);
    G[idx] = (i < N) ? g : 0.0;
  }
}

// balanced base-128 digits of q (|q| < 2^41), most significant first
static void digits6(long long q, signed char* d) {
  for (int p = 5; p >= 0; --p) {
    long long r = q & 127;
    if (r >= 64) r -= 128;
    q = (q - r) >> 7;
    d[p] = (signed char)r;
  }
}

struct Null {
  long long N, ld;
  int d;
  std::vector<long long> V;            // per sample, units 2^-42
  std::vector<long long> X;            // [16][ld] fixed point
  std::vector<unsigned char> vq, dq, xq;   // device images
  double scale[16];
};
static Null make_null(long long N, int d, unsigned long long seed) {
  Null nl;
  nl.N = N;
  nl.ld = (N + 15) / 16 * 16;
  nl.d = d;
  const long long ld = nl.ld;
  nl.V.assign(ld, 0);
  nl.X.assign(16 * ld, 0);
  nl.vq.assign(ld * 8, 0);
  const long long ngroups = (ld + 63) / 64 + 4;  // (four groups of padding: an iteration is fetched as one range)
  const int ncols = d + 2;
  nl.xq.assign(ngroups * 6 * 4 * ncols * 16, 0);
  nl.dq.assign(ngroups * 768, 0);
  for (long long i = 0; i < N; ++i) {
    nl.V[i] = (long long)(mix(seed + i) % (1ull << 40));  // v < 1/4
    signed char dg[6];
    digits6(nl.V[i], dg);
    unsigned char* grp = nl.vq.data() + (size_t)(i >> 2) * 32 + (size_t)(i & 3);
    for (int p = 0; p < 6; ++p) grp[p * 4] = (unsigned char)dg[p];
    {
      const long long g = i >> 6, T = (i >> 4) & 3, q = (i >> 2) & 3, l = i & 3;
      for (int j = 0; j < 3; ++j) {
        unsigned char* e = nl.dq.data() + (size_t)g * 768 + ((q * 3 + j) * 4 + T) * 16 + l;
        e[0] = (unsigned char)dg[2 * j];
        e[4] = (unsigned char)dg[2 * j + 1];
        e[8] = (unsigned char)(signed char)(2 * dg[2 * j]);
        e[12] = (unsigned char)(signed char)(2 * dg[2 * j + 1]);
      }
    }
    for (int k = 0; k < d + 2; ++k) {
      const long long x = (long long)(mix(seed * 77 + i * 16 + k) % (1ull << 41)) - (1ll << 40);
      nl.X[(size_t)k * ld + i] = x;
      digits6(x, dg);
      const long long g = i >> 6, T = (i >> 4) & 3, q = (i >> 2) & 3, l = i & 3;
      for (int p = 0; p < 6; ++p) nl.xq[(((size_t)(g * 6 + p) * 4 + q) * ncols + k) * 16 + T * 4 + l] = (unsigned char)dg[p];
    }
  }
  for (int k = 0; k < 16; ++k) nl.scale[k] = std::ldexp(1.0, -42 + k % 3);
  return nl;
}

struct DevGenes {
  std::vector<GeneDesc> gds;
  GeneDesc* dgd = nullptr;
  double *dG = nullptr, *parts = nullptr, *colstat = nullptr, *bparts = nullptr;
  unsigned* wflags = nullptr;
  unsigned long long* pqw = nullptr;
  size_t gstride = 0, pq_stride = 0;
  int nw = 0, Mp = 0, Cp = 0, ngenes = 0;
  void free_all() {
    hipFree(dG); hipFree(parts); hipFree(colstat); hipFree(bparts); hipFree(wflags); hipFree(pqw); hipFree(dgd);
  }
};
static DevGenes make_genes(long long N, long long ld, int d, int MT, int Mlo, int Mhi, int ngenes, int nw, long long spw,
                           int miss_ppm, bool with_pq, double rate = 0.02) {
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
  D.pq_stride = hcx_pq_entries(D.Mp);
  CK(hipMalloc(&D.pqw, sizeof(unsigned long long) * D.pq_stride * ngenes));
  CK(hipMemset(D.pqw, 0, sizeof(unsigned long long) * D.pq_stride * ngenes));
  D.gds.resize(ngenes);
  for (int g = 0; g < ngenes; ++g) {
    GeneDesc& gd = D.gds[g];
    memset(&gd, 0, sizeof(gd));
    gd.G = D.dG + D.gstride * g;
    const int Mg = Mlo + (g * 7) % (Mhi - Mlo + 1);
    hipLaunchKernelGGL(fill_G, dim3(1024), dim3(256), 0, 0, D.dG + D.gstride * g, ld, N, Mg, 7ull + g, miss_ppm, rate);
    gd.M = Mg; gd.MT = MT; gd.CT = (Mg + d + 1 + 15) / 16; gd.Mp = D.Mp; gd.Cp = 16 * gd.CT;
    gd.n_wparts = nw; gd.steps_per_wpart = (int)spw;
    gd.parts = D.parts + (size_t)g * nw * D.Mp * D.Cp;
    gd.colstat = D.colstat + (size_t)g * nw * kHcColstatRows * D.Mp;
    gd.wflags = D.wflags + (size_t)g * nw;
    gd.bparts = D.bparts + (size_t)g * nw * 2 * (3 + d);
    gd.n_bparts = nw; gd.hc = 1;
    gd.pqw = with_pq ? D.pqw + D.pq_stride * g : nullptr;
    for (int j = 0; j < Mg; ++j)
      if (j % 11 == 3) gd.pflip[j >> 4] |= (unsigned short)(1u << (j & 15));
  }
  CK(hipMalloc(&D.dgd, sizeof(GeneDesc) * ngenes));
  CK(hipMemcpy(D.dgd, D.gds.data(), sizeof(GeneDesc) * ngenes, hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());
  return D;
}

typedef void (*hcx_kernel_t)(const GeneDesc*, NullTileX, long long, long long, int);
static hcx_kernel_t hcx_kernel(int MT) {
  switch (MT) {
    case 1: return gene_suffstat_hcx<1>;
    case 2: return gene_suffstat_hcx<2>;
    case 3: return gene_suffstat_hcx<3>;
    case 4: return gene_suffstat_hcx<4>;
    default: return gene_suffstat_hcx<5>;
  }
}
typedef void (*hcw_kernel_t)(const GeneDesc*, NullTileW, long long, long long, int);
static hcw_kernel_t hcw_kernel(int MT) {
  switch (MT) {
    case 1: return gene_suffstat_hcw<1, 2, 3>;
    case 2: return gene_suffstat_hcw<2, 2, 2>;
    case 3: return gene_suffstat_hcw<3, 2, 2>;
    case 4: return gene_suffstat_hcw<4, 2, 1>;
    default: return gene_suffstat_hcw<5, 1, 1>;
  }
}

static int check() {
  int bad_total = 0;
  const int d = 3;
  const struct { long long N; int M; int miss; int steps; double rate; } cases[] = {
      {49152, 80, 100, 3072, 0.7}, {5000, 80, 3000, 32}, {5000, 65, 0, 48}, {3333, 50, 5000, 16}, {4097, 64, 2000, 32}, {2600, 37, 8000, 64},
      {3000, 20, 3000, 16}, {1000, 9, 20000, 16}, {777, 1, 50000, 16}, {6000, 33, 0, 400}};
  for (const auto& cs : cases) {
    const long long N = cs.N;
    const Null nl = make_null(N, d, 1234 + cs.M);
    const long long ld = nl.ld, nsteps = ld >> 4;
    const int M = cs.M, MT = (M + 15) / 16;
    const long long spw = cs.steps;
    const int nw = (int)((nsteps + spw - 1) / spw);
    DevGenes D = make_genes(N, ld, d, MT, M, M, 1, nw, spw, cs.miss, true, cs.rate > 0 ? cs.rate : 0.02);
    unsigned char *dvq, *dxq;
    CK(hipMalloc(&dvq, nl.dq.size()));
    CK(hipMalloc(&dxq, nl.xq.size()));
    CK(hipMemcpy(dvq, nl.dq.data(), nl.dq.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dxq, nl.xq.data(), nl.xq.size(), hipMemcpyHostToDevice));
    NullTileX nt;
    nt.dq = dvq;
    nt.xq = dxq;
    for (int k = 0; k < 16; ++k) nt.scale[k] = nl.scale[k];
    nt.ncols = d + 2;
    CK(hipMemset(D.parts, 0xff, sizeof(double) * (size_t)nw * D.Mp * D.Cp));
    hipLaunchKernelGGL(hcx_kernel(MT), dim3(nw, 1), dim3(2 * kHcxNW * 64), 0, 0, D.dgd, nt, N, ld, d);
    CK(hipDeviceSynchronize());
    const GeneDesc& gd = D.gds[0];
    const int Mp = D.Mp, Cp = gd.Cp;
    std::vector<double> G((size_t)ld * M), parts((size_t)nw * Mp * Cp), colstat((size_t)nw * kHcColstatRows * Mp),
        bparts((size_t)nw * 2 * (3 + d));
    std::vector<unsigned> wfl(nw);
    std::vector<unsigned long long> pq(D.pq_stride);
    CK(hipMemcpy(G.data(), D.dG, sizeof(double) * G.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(parts.data(), D.parts, sizeof(double) * parts.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(colstat.data(), D.colstat, sizeof(double) * colstat.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(bparts.data(), D.bparts, sizeof(double) * bparts.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(wfl.data(), D.wflags, sizeof(unsigned) * nw, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pq.data(), D.pqw, sizeof(unsigned long long) * pq.size(), hipMemcpyDeviceToHost));
    // ---- host: exact integers -----------------------------------------------------------------------------------
    std::vector<int> H((size_t)N * M), Mk((size_t)N * M);
    long long n_masked = 0;
    for (int j = 0; j < M; ++j)
      for (long long i = 0; i < N; ++i) {
        const double g = G[(size_t)j * ld + i];
        const bool hard = (g == 0.0 || g == 1.0 || g == 2.0);
        H[(size_t)i * M + j] = hard ? (int)g : 0;
        Mk[(size_t)i * M + j] = hard ? 0 : 1;
        n_masked += hard ? 0 : 1;
      }
    int bad = 0;
    auto expect = [&](bool ok, const char* what, int a, int b, double got, double want) {
      if (!ok && bad++ < 8) printf("   MISMATCH %s [%d,%d]: got %.17g want %.17g\n", what, a, b, got, want);
    };
    // Gram and null tiles: sum over parts
    for (int j = 0; j < M; ++j)
      for (int k = j; k < M; ++k) {
        if ((k >> 4) < (j >> 4)) continue;
        long long s = 0;
        for (long long i = 0; i < N; ++i) s += nl.V[i] * H[(size_t)i * M + j] * H[(size_t)i * M + k];
        double got = 0;
        for (int p = 0; p < nw; ++p) got += parts[((size_t)p * Mp + j) * Cp + k];
        expect(got == std::ldexp((double)s, -42), "G'VG", j, k, got, std::ldexp((double)s, -42));
      }
    // (elements of the upper TILE triangle below the diagonal inside a diagonal tile)
    for (int j = 0; j < M; ++j)
      for (int k = (j >> 4) << 4; k < j; ++k) {
        long long s = 0;
        for (long long i = 0; i < N; ++i) s += nl.V[i] * H[(size_t)i * M + j] * H[(size_t)i * M + k];
        double got = 0;
        for (int p = 0; p < nw; ++p) got += parts[((size_t)p * Mp + j) * Cp + k];
        expect(got == std::ldexp((double)s, -42), "G'VG(lower in tile)", j, k, got, std::ldexp((double)s, -42));
      }
    for (int j = 0; j < M; ++j)
      for (int k = 0; k < 16 && M + k < Cp; ++k) {
        long long s = 0;
        for (long long i = 0; i < N; ++i) s += nl.X[(size_t)k * ld + i] * H[(size_t)i * M + j];
        double got = 0;
        for (int p = 0; p < nw; ++p) got += parts[((size_t)p * Mp + j) * Cp + M + k];
        const double want = (k <= d) ? (double)s * nl.scale[k] : 0.0;
        expect(got == want, "H'V[X|res]", j, k, got, want);
      }
    // P, Q, R
    for (int j = 0; j < M; ++j)
      for (int k = 0; k < M; ++k) {
        long long sp = 0, sq = 0;
        for (long long i = 0; i < N; ++i) {
          if (!Mk[(size_t)i * M + j]) continue;
          sp += nl.V[i] * H[(size_t)i * M + k];
          sq += nl.V[i] * Mk[(size_t)i * M + k];
        }
        expect((long long)pq[(size_t)j * Mp + k] == sp, "P", j, k, (double)(long long)pq[(size_t)j * Mp + k], (double)sp);
        if (k >= j) expect((long long)pq[(size_t)Mp * Mp + (size_t)j * Mp + k] == sq, "Q", j, k,
                           (double)(long long)pq[(size_t)Mp * Mp + (size_t)j * Mp + k], (double)sq);
      }
    for (int j = 0; j < M; ++j)
      for (int k = 0; k < 16; ++k) {
        long long s = 0;
        for (long long i = 0; i < N; ++i)
          if (Mk[(size_t)i * M + j]) s += nl.X[(size_t)k * ld + i];
        const long long got = (long long)pq[2 * (size_t)Mp * Mp + (size_t)j * 16 + k];
        expect(got == s, "R", j, k, (double)got, (double)s);
      }
    // column statistics
    for (int j = 0; j < M; ++j) {
      long long s = 0, cm = 0;
      double mn = INFINITY, mx = -INFINITY;
      unsigned long long orb = 0, andb = ~0ull;
      for (long long i = 0; i < N; ++i) {
        if (Mk[(size_t)i * M + j]) {
          ++cm;
          unsigned long long b;
          memcpy(&b, &G[(size_t)j * ld + i], 8);
          orb |= b;
          andb &= b;
        } else {
          s += H[(size_t)i * M + j];
          mn = fmin(mn, H[(size_t)i * M + j]);
          mx = fmax(mx, H[(size_t)i * M + j]);
        }
      }
      double gs = 0, gcm = 0, gmn = INFINITY, gmx = -INFINITY;
      unsigned long long gor = 0, gand = ~0ull;
      for (int p = 0; p < nw; ++p) {
        const double* c = colstat.data() + (size_t)p * kHcColstatRows * Mp;
        gs += c[j];
        gmn = fmin(gmn, c[Mp + j]);
        gmx = fmax(gmx, c[2 * Mp + j]);
        gcm += c[3 * Mp + j];
        gor |= reinterpret_cast<const unsigned long long*>(c)[4 * Mp + j];
        gand &= reinterpret_cast<const unsigned long long*>(c)[5 * Mp + j];
      }
      expect(gs == (double)s, "colsum", j, 0, gs, (double)s);
      expect(gcm == (double)cm, "masked count", j, 0, gcm, (double)cm);
      expect(gmn == mn && gmx == mx, "min/max", j, 0, gmn, mn);
      expect(gor == orb && gand == andb, "OR/AND", j, 0, (double)gor, (double)orb);
    }
    // burden sums
    {
      long long U[2] = {0, 0}, cvc[2] = {0, 0}, cx[2][16] = {{0}}, cnt = 0;
      for (long long i = 0; i < N; ++i) {
        int n = 0;
        for (int j = 0; j < M; ++j) {
          const bool flip = (gd.pflip[j >> 4] >> (j & 15)) & 1;
          const int h = H[(size_t)i * M + j];
          if (Mk[(size_t)i * M + j]) continue;
          n += flip ? (h != 2) : (h != 0);
        }
        const long long c[2] = {n > 0 ? 1 : 0, n};
        cnt += n > 0;
        for (int t = 0; t < 2; ++t) {
          U[t] += c[t] * nl.X[(size_t)d * ld + i];
          cvc[t] += c[t] * c[t] * nl.X[(size_t)(d + 1) * ld + i];
          for (int k = 0; k < d; ++k) cx[t][k] += c[t] * nl.X[(size_t)k * ld + i];
        }
      }
      const int rl = 3 + d;
      for (int t = 0; t < 2; ++t) {
        double gU = 0, gc = 0, gn = 0, gx[16] = {0};
        for (int p = 0; p < nw; ++p) {
          const double* b = bparts.data() + ((size_t)p * 2 + t) * rl;
          gU += b[0];
          gc += b[1];
          gn += b[2];
          for (int k = 0; k < d; ++k) gx[k] += b[3 + k];
        }
        expect(gU == (double)U[t] * nl.scale[d], "burden U", t, 0, gU, (double)U[t] * nl.scale[d]);
        expect(gc == (double)cvc[t] * nl.scale[d + 1], "burden c'Vc", t, 0, gc, (double)cvc[t] * nl.scale[d + 1]);
        expect(gn == (double)cnt, "burden count", t, 0, gn, (double)cnt);
        for (int k = 0; k < d; ++k) expect(gx[k] == (double)cx[t][k] * nl.scale[k], "burden c'VX", t, k, gx[k], (double)cx[t][k] * nl.scale[k]);
      }
    }
    unsigned fl = 0;
    for (int p = 0; p < nw; ++p) fl |= wfl[p];
    expect((fl & 1u) == (n_masked > 0 ? 1u : 0u) && !(fl & 2u), "wflags", 0, 0, fl, n_masked > 0);
    printf("check N=%lld M=%d (MT %d) parts=%d x %lld steps, %lld masked entries: %s\n", N, M, MT, nw, spw, n_masked, bad ? "FAILED" : "ok");
    bad_total += bad;
    D.free_all();
    hipFree(dvq);
    hipFree(dxq);
  }
  return bad_total ? 1 : 0;
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  bool spread = false, miss = false;
  for (int a = 1; a < argc; ++a) {
    if (!strcmp(argv[a], "check")) return check();
    if (!strcmp(argv[a], "spread")) spread = true;
    if (!strcmp(argv[a], "miss")) miss = true;
  }
  const long long N = 200000;
  const int d = 3;
  const Null nl = make_null(N, d, 99);
  const long long ld = nl.ld, nsteps = ld >> 4;
  unsigned char *dvq, *dxq;
  CK(hipMalloc(&dvq, nl.vq.size()));
  CK(hipMalloc(&dxq, nl.xq.size()));
  CK(hipMemcpy(dvq, nl.vq.data(), nl.vq.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dxq, nl.xq.data(), nl.xq.size(), hipMemcpyHostToDevice));
  unsigned char* ddq;
  CK(hipMalloc(&ddq, nl.dq.size()));
  CK(hipMemcpy(ddq, nl.dq.data(), nl.dq.size(), hipMemcpyHostToDevice));
  NullTileX ntx;
  ntx.dq = ddq;
  ntx.xq = dxq;
  for (int k = 0; k < 16; ++k) ntx.scale[k] = nl.scale[k];
  ntx.ncols = d + 2;
  double* dT;
  CK(hipMalloc(&dT, sizeof(double) * ld * (d + 3)));
  CK(hipMemset(dT, 0, sizeof(double) * ld * (d + 3)));
  NullTileW ntw{dT, d + 3, dvq};
  const int Ms[] = {12, 28, 44, 60, 76};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipStream_t bst = 0;
  if (const char* e = getenv("HCX_CUS")) {  // restrict the launches to the LAST k mask bits (the p-value kernel takes the first)
    const int k = atoi(e), ncu = 256, words = 8;
    uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = ncu - k; b < ncu; ++b) m[b / 32] |= 1u << (b % 32);
    CK(hipExtStreamCreateWithCUMask(&bst, words, m));
    printf("launches restricted to %d CUs\n", k);
  }
  for (int Mtop : Ms) {
    const int MT = (Mtop + 15) / 16;
    const int Mlo = spread ? (MT == 2 ? 20 : 16 * (MT - 1) + 1) : Mtop, Mhi = spread ? 16 * MT : Mtop;
    if (spread && MT == 1) continue;
    const int ngenes = spread ? (int)(512.0 * (Mhi - Mlo + 1) / 61.0 + 0.5) : 128;
    for (int variant = 0; variant < 3; ++variant) {  // 0: one-wave kernel, 64 parts; 1: cooperative, 16 parts; 2: cooperative, 32 parts
      if (variant == 0 && miss) continue;            // (the one-wave kernel hands such genes back)
      const int target = variant == 0 ? 64 : (variant == 1 ? 16 : 32), unit = variant == 0 ? kHcStepUnit : kHcxIterSteps;
      long long spw = (nsteps + target - 1) / target;
      spw = (spw + unit - 1) / unit * unit;
      const int nw = (int)((nsteps + spw - 1) / spw);
      DevGenes D = make_genes(N, ld, d, MT, Mlo, Mhi, ngenes, nw, spw, miss ? 1000 : 0, true);
      double sumM = 0;
      for (const GeneDesc& g : D.gds) sumM += g.M;
#ifdef HCX_PROF
      unsigned long long* dprof;
      CK(hipMalloc(&dprof, 32 * 8));
      CK(hipMemset(dprof, 0, 32 * 8));
      for (GeneDesc& g : D.gds) g.dbg_cmc = reinterpret_cast<double*>(dprof);
      CK(hipMemcpy(D.dgd, D.gds.data(), sizeof(GeneDesc) * ngenes, hipMemcpyHostToDevice));
#endif
      auto launch = [&]() {
        if (variant == 0)
          hipLaunchKernelGGL(hcw_kernel(MT), dim3(nw, ngenes), dim3(64), 0, bst, D.dgd, ntw, N, ld, d);
        else
          hipLaunchKernelGGL(hcx_kernel(MT), dim3(nw, ngenes), dim3(2 * kHcxNW * 64), 0, bst, D.dgd, ntx, N, ld, d);
      };
      launch();
      CK(hipDeviceSynchronize());
      const int reps = 5;
      CK(hipEventRecord(e0, bst));
      for (int r = 0; r < reps; ++r) launch();
      CK(hipEventRecord(e1, bst));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = (8.0 * N * (sumM / ngenes) + 8.0 * N * (d + 4)) * ngenes * reps;
      printf("bench M=%d..%d MT=%d %s parts=%d%s: %.3f ms per %d genes, %.2f TB/s algorithmic\n", Mlo, Mhi, MT,
             variant == 0 ? "hcw (one wave) " : "hcx (4 waves)  ", nw, miss ? " 0.1% imputed" : "", ms / reps, ngenes,
             bytes / (ms * 1e-3) / 1e12);
#ifdef HCX_PROF
      if (variant) {
        unsigned long long hp[32];
        CK(hipMemcpy(hp, dprof, sizeof(hp), hipMemcpyDeviceToHost));
        const double nwg = (double)nw * ngenes * (reps + 1), niter = nwg * (double)(spw / kHcxIterSteps);
        for (int w = 0; w < 4; ++w)
          printf("   loader %d cycles per iteration: load %.0f  barrier %.0f  masked %.0f\n", w, hp[w * 4] / niter,
                 hp[w * 4 + 1] / niter, hp[w * 4 + 2] / niter);
        for (int w = 4; w < 8; ++w)
          printf("   tile wave %d cycles per iteration: barrier %.0f  tiles %.0f  masked %.0f\n", w - 4, hp[w * 4] / niter,
                 hp[w * 4 + 1] / niter, hp[w * 4 + 2] / niter);
      }
#endif
      D.free_all();
    }
  }
  return 0;
}
