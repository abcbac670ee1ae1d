// tools/rotgemm_bench.hip — standalone check + micro-benchmark of the int8 rotation GEMM (rvtests_amd/csrc/rot_gemm.hip.h)
//   check: G~ = U'G from the digit planes against a host fp64 product of the dequantised U (exact) and of the float U
//   bench: N x N planes of U against T columns: time per plane pair, effective TOP/s
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rotgemm_bench.hip -o tools/rotgemm_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../rvtests_amd/csrc/rot_gemm.hip.h"
using namespace rvt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

static unsigned long long mix(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
__global__ void fill_i8(signed char* p, long long n, unsigned long long seed, int lo, int hi) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned long long x = (unsigned long long)i * 0x9E3779B97F4A7C15ull + seed;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x ^= x >> 27;
    p[i] = (signed char)(lo + (int)(x % (unsigned long long)(hi - lo + 1)));
  }
}

static void launch_gemm(const signed char* A, const signed char* B, long long ldk, long long kbytes, double* C, long long ldc,
                        int M, int N, const double* cs, double weight, int accumulate) {
  const int nrp = (M + kRotBM - 1) / kRotBM, nct = (N + kRotBN - 1) / kRotBN;
  const int nrpg = (nrp + 31) / 32, nctg = (nct + 7) / 8;
  const long long sets = (long long)nrpg * nctg;
  hipLaunchKernelGGL(rot_gemm_i8_kernel, dim3((unsigned)(sets * 32 * 8)), dim3(kRotThreads), 0, 0, (const int8_t*)A, (const int8_t*)B, ldk,
                     kbytes, C, ldc, M, N, nrp, nct, cs, (const double*)nullptr, weight, accumulate, kbytes, 0LL);
}

int main(int argc, char** argv) {
  const bool bench_only = argc > 1 && !strcmp(argv[1], "bench");
  CK(hipSetDevice(0));
  int fails = 0;

  if (!bench_only) {
    struct Case { int n, M, T, general; };
    const Case cases[] = {{1000, 1000, 200, 0}, {777, 777, 130, 0}, {2048, 2048, 384, 0}, {1500, 1500, 70, 1}, {300, 300, 5, 0}};
    for (const Case& cs : cases) {
      const int n = cs.n, M = cs.M, T = cs.T, PU = kRotPlanesU, PG = cs.general ? kRotPlanesG : 1;
      const long long ldk = (n + 127) / 128 * 128, kbytes = (n + kRotKC - 1) / kRotKC * kRotKC;
      const long long Mpad = (long long)(M + kRotBM - 1) / kRotBM * kRotBM, Tpad = (long long)(T + kRotBN - 1) / kRotBN * kRotBN;
      std::vector<float> U((size_t)n * M);
      std::vector<double> G((size_t)n * T);
      for (size_t i = 0; i < U.size(); ++i) {
        const double u = ((double)(mix(i * 7 + 1) >> 11) / 9007199254740992.0 - 0.5) * 2.0;
        U[i] = (float)(u * ((i % 5 == 0) ? 1e-4 : (i % 3 == 0 ? 0.03 : 1.0)));
      }
      for (size_t i = 0; i < G.size(); ++i) {
        const unsigned long long h = mix(i * 13 + 5);
        G[i] = cs.general ? ((h & 7) == 0 ? 0.37218 * (double)((h >> 8) & 3) : (double)((h >> 3) % 3)) : (double)((h >> 3) % 3);
      }
      float* dUf; double* dG; signed char *dA, *dB; double *dC, *dcs; int *dflag, *dsexp;
      CK(hipMalloc(&dUf, sizeof(float) * U.size()));
      CK(hipMalloc(&dG, sizeof(double) * G.size()));
      CK(hipMalloc(&dA, (size_t)PU * Mpad * ldk));
      CK(hipMalloc(&dB, (size_t)PG * Tpad * ldk));
      CK(hipMalloc(&dC, sizeof(double) * (size_t)M * T));
      CK(hipMalloc(&dcs, sizeof(double) * T));
      CK(hipMalloc(&dflag, sizeof(int)));
      CK(hipMalloc(&dsexp, sizeof(int) * T));
      CK(hipMemset(dA, 0, (size_t)PU * Mpad * ldk));
      CK(hipMemset(dB, 0, (size_t)PG * Tpad * ldk));
      CK(hipMemset(dflag, 0, sizeof(int)));
      CK(hipMemcpy(dUf, U.data(), sizeof(float) * U.size(), hipMemcpyHostToDevice));
      CK(hipMemcpy(dG, G.data(), sizeof(double) * G.size(), hipMemcpyHostToDevice));
      const int sU = 7 * PU - 3;  // |u| < 2
      hipLaunchKernelGGL(rot_quantize_f32_kernel, dim3(1024), dim3(256), 0, 0, dUf, (long long)n, (long long)M, (long long)n, sU, PU,
                         dA, ldk, Mpad * ldk, 0LL, dflag);
      // columns
      std::vector<double> cmax(T);
      double* dmax;
      CK(hipMalloc(&dmax, sizeof(double) * T));
      hipLaunchKernelGGL(rot_colmax_kernel, dim3(T), dim3(256), 0, 0, dG, (long long)n, (long long)n, dmax);
      CK(hipMemcpy(cmax.data(), dmax, sizeof(double) * T, hipMemcpyDeviceToHost));
      std::vector<int> sexp(T, 0);
      std::vector<double> colscale(T);
      bool all_small = true;
      for (int j = 0; j < T; ++j) all_small = all_small && cmax[j] >= 0 && cmax[j] <= 127.0;
      if ((PG == 1) != all_small) { printf("plane decision mismatch (PG=%d all_small=%d)\n", PG, (int)all_small); ++fails; }
      for (int j = 0; j < T; ++j) {
        const double mx = cmax[j] < 0 ? -cmax[j] - 1.0 : cmax[j];
        sexp[j] = (PG == 1) ? 0 : (mx > 0 ? 7 * PG - 3 - ilogb(mx) : 0);
        colscale[j] = ldexp(1.0, -(sU + sexp[j]));
      }
      CK(hipMemcpy(dsexp, sexp.data(), sizeof(int) * T, hipMemcpyHostToDevice));
      CK(hipMemcpy(dcs, colscale.data(), sizeof(double) * T, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(rot_quantize_f64_kernel, dim3(1024), dim3(256), 0, 0, dG, (long long)n, (long long)T, (long long)n, dsexp, PG, dB,
                         ldk, Tpad * ldk);
      int first = 1;
      for (int s = 0; s <= (PU - 1) + (PG - 1); ++s)          // least significant plane pairs first
        for (int p = 0; p < PU; ++p) {
          const int q = s - p;
          if (q < 0 || q >= PG) continue;
          launch_gemm(dA + (size_t)p * Mpad * ldk, dB + (size_t)q * Tpad * ldk, ldk, kbytes, dC, M, M, T, dcs, ldexp(1.0, 7 * (p + q)), first ? 0 : 1);
          first = 0;
        }
      CK(hipDeviceSynchronize());
      int hflag = 0;
      CK(hipMemcpy(&hflag, dflag, sizeof(int), hipMemcpyDeviceToHost));
      std::vector<double> C((size_t)M * T);
      CK(hipMemcpy(C.data(), dC, sizeof(double) * C.size(), hipMemcpyDeviceToHost));
      double worst_q = 0, worst_t = 0, scale_ref = 0;
      for (int j = 0; j < T; ++j)
        for (int k = 0; k < M; ++k) {
          double sq = 0, st = 0, sa = 0;
          for (int i = 0; i < n; ++i) {
            const double u = (double)U[(size_t)k * n + i], g = G[(size_t)j * n + i];
            const double uq = (double)llrint(u * ldexp(1.0, sU)) * ldexp(1.0, -sU);
            const double gq = (PG == 1) ? g : (double)llrint(ldexp(g, sexp[j])) * ldexp(1.0, -sexp[j]);
            sq += uq * gq;
            st += u * g;
            sa += fabs(u * g);
          }
          worst_q = fmax(worst_q, fabs(C[(size_t)j * M + k] - sq) / fmax(sa, 1e-300));
          worst_t = fmax(worst_t, fabs(C[(size_t)j * M + k] - st));
          scale_ref = fmax(scale_ref, fabs(st));
        }
      const bool ok = hflag == 0 && worst_q < 1e-14 && worst_t < 1e-9 * scale_ref;
      printf("check n=%d M=%d T=%d planesG=%d: vs dequantised product rel %.3g, vs float U abs %.3g (max |G~| %.3g)  %s\n", n, M, T, PG,
             worst_q, worst_t, scale_ref, ok ? "OK" : "FAIL");
      if (!ok) ++fails;
      hipFree(dUf); hipFree(dG); hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dcs); hipFree(dflag); hipFree(dsexp); hipFree(dmax);
    }
  }
  // ---- bench ------------------------------------------------------------------------------------------------------------
  {
    const long long sizes[] = {20000, 40000, 100000};
    for (long long n : sizes) {
      const int T = 3840;
      const long long ldk = (n + 127) / 128 * 128, kbytes = (n + kRotKC - 1) / kRotKC * kRotKC;
      const long long Mpad = (n + kRotBM - 1) / kRotBM * kRotBM, Tpad = (long long)(T + kRotBN - 1) / kRotBN * kRotBN;
      signed char *dA, *dB; double *dC, *dcs;
      if (hipMalloc(&dA, (size_t)Mpad * ldk) != hipSuccess) { printf("skip n=%lld (no memory)\n", n); continue; }
      CK(hipMalloc(&dB, (size_t)Tpad * ldk));
      CK(hipMalloc(&dC, sizeof(double) * (size_t)n * T));
      CK(hipMalloc(&dcs, sizeof(double) * T));
      hipLaunchKernelGGL(fill_i8, dim3(4096), dim3(256), 0, 0, dA, Mpad * ldk, 1ull, -64, 63);
      hipLaunchKernelGGL(fill_i8, dim3(4096), dim3(256), 0, 0, dB, Tpad * ldk, 2ull, 0, 2);
      std::vector<double> one(T, 1.0);
      CK(hipMemcpy(dcs, one.data(), sizeof(double) * T, hipMemcpyHostToDevice));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      launch_gemm(dA, dB, ldk, kbytes, dC, n, (int)n, T, dcs, 1.0, 0);
      CK(hipDeviceSynchronize());
      const int reps = 3;
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) launch_gemm(dA, dB, ldk, kbytes, dC, n, (int)n, T, dcs, 1.0, 1);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double ops = 2.0 * (double)n * (double)n * T;
      printf("bench n=%lld T=%d: %.2f ms per plane pair, %.2f POP/s; 6 planes = %.1f ms per batch of 128 genes (M=30) = %.0f gene-sets/s\n", n, T,
             ms / reps, ops / (ms / reps * 1e-3) / 1e15, 6 * ms / reps, 128.0 / (6 * ms / reps * 1e-3));
      hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dcs);
    }
  }
  return fails ? 1 : 0;
}
