// tools/gemm64_bench.hip — standalone check + micro-benchmark of the fp64 band product (rvtests_amd/csrc/gemm_f64.hip.h)
//   check: C = A' D B on small integer-valued operands (exact in fp64 whatever the summation order) against a host product,
//          with and without weights, symmetric and rectangular, K not a multiple of the chunk, several K slices
//   bench: N x V block, the upper triangle of G'G: time per launch, TFLOP/s of the band (2 N V^2 / 2) and of the tiles computed
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm64_bench.hip -o tools/gemm64_bench
// usage: tools/gemm64_bench [check|bench] [N] [V] [slices]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../rvtests_amd/csrc/gemm_f64.hip.h"
using namespace rvt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

static unsigned long long mix(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
__global__ void fill_f64(double* p, long long n, long long ld, long long rows, unsigned long long seed) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned long long x = (unsigned long long)i * 0x9E3779B97F4A7C15ull + seed;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x ^= x >> 27;
    p[i] = (i % ld) < rows ? (double)(x % 3) : 0.0;
  }
}

// slices = 0: the engine's heuristic
static int launch(const double* A, long long lda, int M, const double* B, long long ldb, int Nb, const double* B2, long long ldb2,
                  int Nb2, const double* w, long long N, double* C, long long ldc, double* part, int symmetric, int slices_in,
                  int halo = -1, int ring = 0, int col0 = 0) {
  const int Ntot = Nb + Nb2;
  int nct = 0;
  const int n_tiles = gemm_f64_tiles(M, Ntot, symmetric != 0, &nct, halo);
  const long long chunks = (N + kGemmKC - 1) / kGemmKC;
  long long slices = slices_in > 0 ? std::min(slices_in, 64) : gemm_f64_slices(n_tiles, chunks);
  const long long kslice = ((chunks + slices - 1) / slices) * kGemmKC;
  slices = (N + kslice - 1) / kslice;
  const long long c_slice = ldc * Ntot;
  const long long groups = (slices + 7) / 8;
  hipLaunchKernelGGL((gemm_tn_f64_kernel<3>), dim3((unsigned)(8 * (long long)n_tiles * groups)),
                     dim3(kGemmThreads), 0, 0, A, lda, M, B, ldb, Nb, B2 ? B2 : B, B2 ? ldb2 : ldb, Nb2, w, N, kslice, (int)slices,
                     slices > 1 ? part : C, ldc, slices > 1 ? c_slice : 0LL, n_tiles, nct, symmetric, halo, ring, col0);
  if (slices > 1)
    hipLaunchKernelGGL(rot_reduce_slices_kernel, dim3(1024), dim3(256), 0, 0, part, ldc, (long long)M, (long long)Ntot, c_slice,
                       (int)slices, C, 0);
  return (int)slices;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "check";
  CK(hipSetDevice(0));
  if (!strcmp(mode, "check")) {
    // halo >= 0: only the band j - m <= halo of a symmetric product is computed (and checked); ring > 0: the columns live in a
    // ring of `ring` physical columns, logical column l at physical (col0 + l) mod ring (MetaCov's circular window)
    struct Case { long long N; int M, Nb, Nb2, sym, weighted, slices, halo, ring, col0; };
    const Case cases[] = {{1000, 300, 300, 0, 1, 0, 0, -1, 0, 0}, {5003, 600, 600, 0, 1, 1, 3, -1, 0, 0}, {777, 40, 130, 3, 0, 0, 1, -1, 0, 0},
                          {4096, 257, 129, 2, 0, 1, 5, -1, 0, 0}, {33, 16, 16, 0, 1, 0, 1, -1, 0, 0}, {20000, 1024, 1024, 0, 1, 0, 0, -1, 0, 0},
                          {3001, 900, 900, 0, 1, 0, 0, 100, 0, 0}, {2000, 700, 700, 0, 1, 1, 2, 300, 1000, 650},
                          {1500, 520, 520, 0, 1, 0, 0, 0, 520, 519}};
    int fails = 0;
    for (const Case& cs : cases) {
      const long long N = cs.N, ld = (N + 15) / 16 * 16;
      const int M = cs.M, Nb = cs.sym ? cs.M : cs.Nb, Nb2 = cs.Nb2, Ntot = Nb + Nb2;
      const int physM = cs.ring > 0 ? cs.ring : M;
      auto phys = [&](int l) { return cs.ring > 0 ? (cs.col0 + l) % cs.ring : l; };
      std::vector<double> hA((size_t)ld * physM, 0.0), hB((size_t)ld * Nb, 0.0), hB2((size_t)ld * std::max(Nb2, 1), 0.0), hw(ld, 0.0);
      for (int j = 0; j < M; ++j)
        for (long long i = 0; i < N; ++i) hA[(size_t)phys(j) * ld + i] = (double)(mix(j * 1000003ull + i) % 3);
      if (cs.ring > 0)   // (the physical columns outside the window hold something else)
        for (int p = 0; p < physM; ++p) {
          bool used = false;
          for (int j = 0; j < M && !used; ++j) used = phys(j) == p;
          if (!used)
            for (long long i = 0; i < N; ++i) hA[(size_t)p * ld + i] = 7.0;
        }
      if (cs.sym) hB = hA;
      else
        for (int j = 0; j < Nb; ++j)
          for (long long i = 0; i < N; ++i) hB[(size_t)j * ld + i] = (double)(mix(77 + j * 999983ull + i) % 5) - 2.0;
      for (int j = 0; j < Nb2; ++j)
        for (long long i = 0; i < N; ++i) hB2[(size_t)j * ld + i] = (double)(mix(5 + j * 7919ull + i) % 7) - 3.0;
      for (long long i = 0; i < N; ++i) hw[i] = 0.25 * (double)(1 + mix(i + 99) % 4);  // dyadic: products stay exact
      double *dA, *dB, *dB2, *dw, *dC, *dP;
      CK(hipMalloc(&dA, hA.size() * 8)); CK(hipMalloc(&dB, hB.size() * 8)); CK(hipMalloc(&dB2, hB2.size() * 8)); CK(hipMalloc(&dw, hw.size() * 8));
      CK(hipMalloc(&dC, (size_t)M * Ntot * 8)); CK(hipMalloc(&dP, (size_t)M * Ntot * 8 * 64));
      CK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice));
      CK(hipMemcpy(dB2, hB2.data(), hB2.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 8, hipMemcpyHostToDevice));
      CK(hipMemset(dC, 0xff, (size_t)M * Ntot * 8));
      const int sl = launch(dA, ld, M, cs.sym ? dA : dB, ld, Nb, Nb2 ? dB2 : nullptr, ld, Nb2, cs.weighted ? dw : nullptr, N, dC, M, dP, cs.sym, cs.slices,
                            cs.halo, cs.ring, cs.col0);
      CK(hipDeviceSynchronize());
      std::vector<double> hC((size_t)M * Ntot);
      CK(hipMemcpy(hC.data(), dC, hC.size() * 8, hipMemcpyDeviceToHost));
      long long bad = 0, checked = 0;
      const int step = (M * (long long)Ntot > 200000) ? 7 : 1;
      for (int m = 0; m < M; m += step)
        for (int j = 0; j < Ntot; j += 1) {
          if (cs.sym && j < m) continue;  // below the diagonal: unspecified
          if (cs.halo >= 0 && j - m > cs.halo) continue;  // outside the band: unspecified
          const double* b = j < Nb ? &hB[(size_t)(cs.sym ? phys(j) : j) * ld] : &hB2[(size_t)(j - Nb) * ld];
          const double* a = &hA[(size_t)phys(m) * ld];
          double s = 0.0;
          for (long long i = 0; i < N; ++i) s += a[i] * (cs.weighted ? hw[i] : 1.0) * b[i];
          ++checked;
          if (s != hC[(size_t)j * M + m]) {
            if (bad < 5) fprintf(stderr, "  mismatch m=%d j=%d got %.17g want %.17g\n", m, j, hC[(size_t)j * M + m], s);
            ++bad;
          }
        }
      printf("check N=%lld M=%d Nb=%d+%d sym=%d weighted=%d slices=%d halo=%d ring=%d col0=%d: %lld / %lld wrong\n", N, M, Nb, Nb2, cs.sym,
             cs.weighted, sl, cs.halo, cs.ring, cs.col0, bad, checked);
      fails += bad != 0;
      hipFree(dA); hipFree(dB); hipFree(dB2); hipFree(dw); hipFree(dC); hipFree(dP);
    }
    printf(fails ? "FAILED\n" : "all checks passed\n");
    return fails ? 1 : 0;
  }
  const long long N = argc > 2 ? atoll(argv[2]) : 500000, ld = (N + 15) / 16 * 16;
  const int V = argc > 3 ? atoi(argv[3]) : 1024;
  const int slices_in = argc > 4 ? atoi(argv[4]) : 0;
  const int sym = strcmp(mode, "full") ? 1 : 0;  // "full": every tile (occupancy probes)
  double *dG, *dC, *dP;
  CK(hipMalloc(&dG, (size_t)ld * V * 8)); CK(hipMalloc(&dC, (size_t)V * V * 8)); CK(hipMalloc(&dP, (size_t)V * V * 8 * 64));
  hipLaunchKernelGGL(fill_f64, dim3(4096), dim3(256), 0, 0, dG, ld * V, ld, N, 12345ull);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int sl = 0;
  for (int rep = 0; rep < 2; ++rep) sl = launch(dG, ld, V, dG, ld, V, nullptr, 0, 0, nullptr, N, dC, V, dP, sym, slices_in);
  CK(hipDeviceSynchronize());
  const int reps = 5;
  CK(hipEventRecord(e0));
  for (int rep = 0; rep < reps; ++rep) launch(dG, ld, V, dG, ld, V, nullptr, 0, 0, nullptr, N, dC, V, dP, sym, slices_in);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const int nrp = (V + kGemmBM - 1) / kGemmBM, nct = (V + kGemmBN - 1) / kGemmBN;
  int active = 0;
  for (int rp = 0; rp < nrp; ++rp)
    for (int ct = 0; ct < nct; ++ct) active += (!sym || (long long)ct * kGemmBN + kGemmBN > (long long)rp * kGemmBM);
  printf("{\"N\": %lld, \"V\": %d, \"slices\": %d, \"active_tiles\": %d, \"ms\": %.3f, \"band_TFLOPs\": %.2f, \"computed_TFLOPs\": %.2f}\n", N, V, sl,
         active, ms, 2.0 * N * V * (V / 2.0) / ms / 1e9, 2.0 * N * (double)active * kGemmBM * kGemmBN / ms / 1e9);
  return 0;
}
