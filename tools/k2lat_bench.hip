// tools/k2lat_bench.hip — micro-benchmark of the lattice-dosage sufficient-statistics kernel
// (rvtests_amd/csrc/suffstat_lat.hip.h) outside the engine: N = 500 000, dosages K / 1000, candidate
// (ring depth, waves per SIMD) configurations; algorithmic TB/s (8 N M + 8 N (d + 2) bytes per gene).
// Results are checked by the engine's tests (tests/test_gpu_lattice.py), not here.
// build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/k2lat_bench.hip -o tools/k2lat_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../rvtests_amd/csrc/suffstat_lat.hip.h"

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

__global__ void fill_G(double* G, long long total, unsigned long long seed) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const unsigned long long h = mix(seed ^ (unsigned long long)idx * 0xD1B54A32D192ED03ull);
    const double a = (double)(h >> 40) * (1.0 / 16777216.0), b = (double)((h >> 16) & 0xffffff) * (1.0 / 16777216.0);
    // imputed dosages of a rare variant: mostly small, a few near 1 and 2
    const unsigned K = (a < 0.02 ? 1000u : 0u) + (b < 0.02 ? 1000u : 0u);
    const unsigned jit = (unsigned)((h >> 3) & 0x3ff) % 61u;  // 0 .. 60
    const unsigned Kj = (h & 4) ? K + jit : (K >= jit ? K - jit : K + jit);
    G[idx] = (double)(Kj > 2000u ? 2000u : Kj) / 1000.0;
  }
}
__global__ void fill_bytes(unsigned* p, long long n, unsigned long long seed) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x)
    p[idx] = (unsigned)mix(seed + idx) & 0x3f3f3f3fu;
}
__global__ void fill_null(double* T, long long total, unsigned long long seed) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x)
    T[idx] = (double)(mix(seed + idx) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
}

typedef void (*hcw_kernel_t)(const GeneDesc*, NullTile, LatParam, long long, long long, int);
struct Cfg {
  int MT, depth, waves;
  hcw_kernel_t k;
};
#define CFG(mt, dp, w) {mt, dp, w, gene_suffstat_lat<mt, dp, w>}
static const Cfg kCfgs[] = {
#ifdef HCW_CFGS
    HCW_CFGS
#else
    CFG(1, 2, 4), CFG(2, 2, 3), CFG(2, 2, 2), CFG(3, 2, 2), CFG(3, 1, 2), CFG(4, 2, 2), CFG(4, 1, 2), CFG(4, 2, 1), CFG(5, 1, 1), CFG(5, 2, 1),
#endif
};

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  const long long N = 500000, ld = (N + 15) / 16 * 16;
  const int d = 3;
  const long long nsteps = ld >> 4;
  long long spw = (nsteps + 63) / 64;
  spw = (spw + kHcStepUnit - 1) / kHcStepUnit * kHcStepUnit;
  if (spw > kHcMaxSteps) spw = kHcMaxSteps;
  const int nw = (int)((nsteps + spw - 1) / spw);
  double* dT;
  CK(hipMalloc(&dT, sizeof(double) * ld * (d + 2)));
  hipLaunchKernelGGL(fill_null, dim3(1024), dim3(256), 0, 0, dT, ld * (d + 2), 99ull);
  NullTile nt{dT, d + 2};
  const LatParam lpar{1000.0};
  // default: 128 genes of one width near the top of each class; "spread" as first argument: the widths of a class as a
  // 512-gene batch with M ~ U{20..80} holds them (every width of the class in turn, 134 genes for the full classes)
  const bool spread = argc > 1 && !strcmp(argv[1], "spread");
  const int Ms[] = {12, 28, 44, 60, 76};
  for (int Mtop : Ms) {
    const int MT = (Mtop + 15) / 16;
    const int Mlo = spread ? (MT == 2 ? 20 : 16 * (MT - 1) + 1) : Mtop, Mhi = spread ? 16 * MT : Mtop;
    if (spread && MT == 1) continue;
    const int ngenes = spread ? (int)(512.0 * (Mhi - Mlo + 1) / 61.0 + 0.5) * 0 + (int)(220.0 * (Mhi - Mlo + 1) / 61.0 + 0.5) : 56;  // (N = 500 000: 15-45 GB per class)
    const int M = Mhi;  // (allocation width)
    const int CTmax = (Mhi + d + 1 + 15) / 16, Mp = 16 * MT, Cp = 16 * CTmax;
    double sumM = 0;
    double* dG;
    const size_t gstride = (size_t)ld * M;
    CK(hipMalloc(&dG, sizeof(double) * gstride * ngenes));
    hipLaunchKernelGGL(fill_G, dim3(4096), dim3(256), 0, 0, dG, (long long)(gstride * ngenes), 7ull);
    double *parts, *colstat, *bparts;
    CK(hipMalloc(&parts, sizeof(double) * (size_t)ngenes * nw * Mp * Cp));
    CK(hipMalloc(&colstat, sizeof(double) * (size_t)ngenes * nw * kHcColstatRows * Mp));
    unsigned* wflags;
    CK(hipMalloc(&wflags, sizeof(unsigned) * (size_t)ngenes * nw));
    CK(hipMalloc(&bparts, sizeof(double) * (size_t)ngenes * nw * 2 * (3 + d)));
    std::vector<GeneDesc> gds(ngenes);
    for (int g = 0; g < ngenes; ++g) {
      GeneDesc& gd = gds[g];
      memset(&gd, 0, sizeof(gd));
      gd.G = dG + gstride * g;
      const int Mg = Mlo + (g * 7) % (Mhi - Mlo + 1);
      sumM += Mg;
      gd.M = Mg; gd.MT = MT; gd.CT = (Mg + d + 1 + 15) / 16; gd.Mp = Mp; gd.Cp = 16 * gd.CT;
      gd.n_wparts = nw; gd.steps_per_wpart = (int)spw;
      gd.parts = parts + (size_t)g * nw * Mp * Cp;
      gd.colstat = colstat + (size_t)g * nw * kHcColstatRows * Mp;
      gd.wflags = wflags + (size_t)g * nw;
      gd.bparts = bparts + (size_t)g * nw * 2 * (3 + d);
      gd.n_bparts = nw; gd.hc = 1;
    }
    GeneDesc* dgd;
    CK(hipMalloc(&dgd, sizeof(GeneDesc) * ngenes));
    CK(hipMemcpy(dgd, gds.data(), sizeof(GeneDesc) * ngenes, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Cfg& cf : kCfgs) {
      if (cf.MT != MT) continue;
      auto launch = [&]() { hipLaunchKernelGGL(cf.k, dim3(nw, ngenes), dim3(64), 0, 0, dgd, nt, lpar, N, ld, d); };
      launch();
      CK(hipDeviceSynchronize());
      const int reps = 5;
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = (8.0 * N * (sumM / ngenes) + 8.0 * N * (d + 2)) * ngenes * reps;
      printf("bench M=%d..%d MT=%d depth=%d waves=%d: %.3f ms per %d genes, %.2f TB/s algorithmic\n", Mlo, Mhi, MT, cf.depth, cf.waves,
             ms / reps, ngenes, bytes / (ms * 1e-3) / 1e12);
    }
    {
      std::vector<unsigned> wf((size_t)nw);
      CK(hipMemcpy(wf.data(), wflags, sizeof(unsigned) * nw, hipMemcpyDeviceToHost));
      unsigned any = 0;
      for (unsigned x : wf) any |= x;
      if (any) printf("  (gene 0 flagged off-lattice: %u)\n", any);
    }
    CK(hipFree(dG)); CK(hipFree(parts)); CK(hipFree(colstat)); CK(hipFree(bparts)); CK(hipFree(dgd)); CK(hipFree(wflags));
  }
  return 0;
}
