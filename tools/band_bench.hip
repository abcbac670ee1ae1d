// tools/band_bench.hip — standalone check + micro-benchmark of the band product of MetaCov's sliding window
// (rvtests_amd/csrc/band_gemm.hip.h): the int8 kernel (v_mfma_i32_32x32x32_i8 on one byte per genotype) against the MXFP4
// kernel (v_mfma_scale_f32_32x32x64_f8f6f4 on E2M1 codes, two genotypes per byte, unit block scales, fp32 accumulation).
//   check: random hard calls 0 / 1 / 2, the window on a ring that wraps; every entry of the band from both kernels must be the
//          same integer, and a sample of entries is compared with a plain dot product of the int8 columns
//   bench: time per launch and POP/s on the tiles computed / on the band printed
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/band_bench.hip -o tools/band_bench
// usage: tools/band_bench [check|bench] [N] [H] [halo] [slices]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../rvtests_amd/csrc/band_gemm.hip.h"
using namespace rvt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

// hard calls with a per-column allele frequency; pad rows [N, ldk) zero
__global__ void fill_i8(int8_t* p, long long cols, long long ldk, long long N, unsigned long long seed) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < cols * ldk; i += (long long)gridDim.x * blockDim.x) {
    const long long c = i / ldk, r = i % ldk;
    unsigned long long x = (unsigned long long)i * 0x9E3779B97F4A7C15ull + seed;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
    const unsigned thr = 40 + (unsigned)((c * 2654435761ull) % 600);   // of 1024: P(allele)
    const int g = ((x & 1023) < thr) + (((x >> 10) & 1023) < thr);
    p[i] = r < N ? (int8_t)g : 0;
  }
}
// int8 0 / 1 / 2 -> E2M1 codes 0x0 / 0x2 / 0x4, sample 2 i in the low nibble of byte i
__global__ void pack_fp4(const int8_t* src, long long cols, long long ldk, uint8_t* dst, long long ldk4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < cols * ldk4; i += (long long)gridDim.x * blockDim.x) {
    const long long c = i / ldk4, b = i % ldk4;
    const int g0 = 2 * b < ldk ? src[c * ldk + 2 * b] : 0, g1 = 2 * b + 1 < ldk ? src[c * ldk + 2 * b + 1] : 0;
    dst[i] = (uint8_t)((g0 << 1) | (g1 << 5));
  }
}
// S[h][t] = sum over slices of the partial tiles (the indexing of band_finish_i32_kernel)
__global__ void band_sum(const int* part, int n_slices, int n_tiles, int H, int W, int halo, long long* S) {
  const int h = blockIdx.x;
  int tile0 = 0;
  for (int rp = 0; rp < (h >> 8); ++rp) tile0 += band_panel_tiles(rp, W, halo);
  for (int t = threadIdx.x; t <= halo; t += blockDim.x) {
    const int j = h + t;
    long long s = -1;
    if (j < W) {
      const int tile = tile0 + (j >> 8) - (h >> 8);
      const int* p = part + ((long long)tile << 16) + (h & 255) * kBandBT + (j & 255);
      s = 0;
      for (int sl = 0; sl < n_slices; ++sl) s += p[((long long)sl * n_tiles) << 16];
    }
    S[(long long)h * (halo + 1) + t] = s;
  }
}
__global__ void count_diff(const long long* a, const long long* b, long long n, unsigned long long* bad) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    if (a[i] != b[i]) atomicAdd(bad, 1ull);
}
// plain dot products of sampled pairs: one workgroup per sample
__global__ void dot_ref(const int8_t* R, long long ldk, long long N, int ring, int col0, const int* hs, const int* ts, long long* out) {
  __shared__ long long red[256];
  long long a = col0 + hs[blockIdx.x], b = col0 + hs[blockIdx.x] + ts[blockIdx.x];
  if (ring > 0) { a %= ring; b %= ring; }
  long long s = 0;
  for (long long i = threadIdx.x; i < N; i += 256) s += (long long)R[a * ldk + i] * R[b * ldk + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

struct Plan { int n_tiles; long long nsl, kslice, kbytes; };
static Plan plan(int H, int W, int halo, long long kbytes, int slices_in, long long max_kslice) {
  Plan p;
  p.n_tiles = band_tiles(H, W, halo);
  p.kbytes = kbytes;
  const long long chunks = kbytes / kRotKC;
  p.nsl = slices_in > 0 ? slices_in : band_slices(p.n_tiles, chunks, (size_t)3 << 30);
  p.kslice = ((chunks + p.nsl - 1) / p.nsl) * kRotKC;
  if (p.kslice > max_kslice) p.kslice = max_kslice / kRotKC * kRotKC;
  p.nsl = (kbytes + p.kslice - 1) / p.kslice;
  return p;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "check";
  CK(hipSetDevice(0));
  const bool check = !strcmp(mode, "check");
  struct Case { long long N; int H, halo, ring, col0, slices; };
  std::vector<Case> cases;
  if (check) {
    cases = {{9100, 700, 300, 1100, 900, 0}, {40000, 1024, 1000, 2304, 2000, 0}, {300000, 512, 200, 0, 0, 0}, {1000, 300, 0, 0, 5, 1},
             {77777, 260, 77, 400, 399, 3}};
  } else {
    cases = {{argc > 2 ? atoll(argv[2]) : 500000, argc > 3 ? atoi(argv[3]) : 1024, argc > 4 ? atoi(argv[4]) : 1000, 0, 0,
              argc > 5 ? atoi(argv[5]) : 0}};
  }
  int fails = 0;
  for (const Case& cs : cases) {
    const long long N = cs.N, ldk = (N + 127) / 128 * 128, ldk4 = ((N + 1) / 2 + 127) / 128 * 128;
    const int H = cs.H, halo = cs.halo, W = H + halo, ring = cs.ring, col0 = cs.col0;
    const long long cols = (ring > 0 ? ring : col0 + W);
    int8_t* R8; uint8_t* R4; int* part; long long *S8, *S4; unsigned long long* d_bad;
    CK(hipMalloc(&R8, (size_t)cols * ldk)); CK(hipMalloc(&R4, (size_t)cols * ldk4));
    hipLaunchKernelGGL(fill_i8, dim3(4096), dim3(256), 0, 0, R8, cols, ldk, N, 4242ull + (unsigned long long)N);
    hipLaunchKernelGGL(pack_fp4, dim3(4096), dim3(256), 0, 0, R8, cols, ldk, R4, ldk4);
    // int8: a slice's sums stay below 2^31 (4 N); fp4: below 2^24 -> at most 2^22 samples = 2^21 bytes per slice
    const Plan p8 = plan(H, W, halo, ldk, cs.slices, 1LL << 40), p4 = plan(H, W, halo, ldk4, cs.slices, 1LL << 21);
    const size_t nt = (size_t)p8.n_tiles, pbytes = sizeof(int) * nt * (size_t)std::max(p8.nsl, p4.nsl) * kBandBT * kBandBT;
    CK(hipMalloc(&part, pbytes));
    const size_t nS = (size_t)H * (halo + 1);
    CK(hipMalloc(&S8, nS * 8)); CK(hipMalloc(&S4, nS * 8)); CK(hipMalloc(&d_bad, 8)); CK(hipMemset(d_bad, 0, 8));
    auto run8 = [&]() {
      hipLaunchKernelGGL(band_gemm_i8, dim3((unsigned)(8 * (long long)p8.n_tiles * ((p8.nsl + 7) / 8))), dim3(kBandThreads), 0, 0, R8, R8, ldk,
                         ring, col0, H, W, halo, p8.kbytes, p8.kslice, (int)p8.nsl, p8.n_tiles, part);
    };
    auto run4 = [&]() {
      hipLaunchKernelGGL(band_gemm_fp4, dim3((unsigned)(8 * (long long)p4.n_tiles * ((p4.nsl + 7) / 8))), dim3(kBandThreads), 0, 0,
                         (const int8_t*)R4, (const int8_t*)R4, ldk4, ring, col0, H, W, halo, p4.kbytes, p4.kslice, (int)p4.nsl, p4.n_tiles, part);
    };
    run8();
    hipLaunchKernelGGL(band_sum, dim3((unsigned)H), dim3(256), 0, 0, part, (int)p8.nsl, p8.n_tiles, H, W, halo, S8);
    CK(hipDeviceSynchronize());
    CK(hipMemset(part, 0xff, pbytes));
    run4();
    hipLaunchKernelGGL(band_sum, dim3((unsigned)H), dim3(256), 0, 0, part, (int)p4.nsl, p4.n_tiles, H, W, halo, S4);
    hipLaunchKernelGGL(count_diff, dim3(1024), dim3(256), 0, 0, S8, S4, (long long)nS, d_bad);
    CK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    // a sample of entries against plain dot products
    const int ns = 512;
    std::vector<int> hs(ns), ts(ns);
    for (int i = 0; i < ns; ++i) { hs[i] = (int)((i * 7919ull) % H); ts[i] = (int)((i * 104729ull) % (halo + 1)); }
    hs[0] = H - 1; ts[0] = halo; hs[1] = 0; ts[1] = 0;
    int *dh, *dt; long long* dref;
    CK(hipMalloc(&dh, ns * 4)); CK(hipMalloc(&dt, ns * 4)); CK(hipMalloc(&dref, ns * 8));
    CK(hipMemcpy(dh, hs.data(), ns * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, ts.data(), ns * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dot_ref, dim3(ns), dim3(256), 0, 0, R8, ldk, N, ring, col0, dh, dt, dref);
    std::vector<long long> ref(ns), h8(nS), h4(nS);
    CK(hipMemcpy(ref.data(), dref, ns * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h8.data(), S8, nS * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h4.data(), S4, nS * 8, hipMemcpyDeviceToHost));
    long long bad_ref8 = 0, bad_ref4 = 0;
    for (int i = 0; i < ns; ++i) {
      const size_t at = (size_t)hs[i] * (halo + 1) + ts[i];
      bad_ref8 += h8[at] != ref[i];
      bad_ref4 += h4[at] != ref[i];
    }
    printf("check N=%lld H=%d halo=%d ring=%d col0=%d tiles=%d slices int8=%lld fp4=%lld: fp4 != int8 in %llu / %zu entries; vs dot products: int8 %lld, fp4 %lld of %d wrong\n",
           N, H, halo, ring, col0, p8.n_tiles, p8.nsl, p4.nsl, bad, nS, bad_ref8, bad_ref4, ns);
    fails += (bad != 0) + (bad_ref8 != 0) + (bad_ref4 != 0);
    if (!check) {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int reps = 10;
      float ms8 = 0, ms4 = 0;
      run8(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) run8(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms8, e0, e1)); ms8 /= reps;
      run4(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) run4(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms4, e0, e1)); ms4 /= reps;
      const double tile_ops = 2.0 * N * (double)p8.n_tiles * kBandBT * kBandBT, band_ops = 2.0 * N * (double)H * (halo + 1);
      printf("{\"N\": %lld, \"H\": %d, \"halo\": %d, \"tiles\": %d, \"int8\": {\"slices\": %lld, \"ms\": %.3f, \"tile_POPs\": %.3f, \"band_POPs\": %.3f}, "
             "\"fp4\": {\"slices\": %lld, \"ms\": %.3f, \"tile_POPs\": %.3f, \"band_POPs\": %.3f}}\n",
             N, H, halo, p8.n_tiles, p8.nsl, ms8, tile_ops / ms8 / 1e12, band_ops / ms8 / 1e12, p4.nsl, ms4, tile_ops / ms4 / 1e12, band_ops / ms4 / 1e12);
    }
    CK(hipFree(R8)); CK(hipFree(R4)); CK(hipFree(part)); CK(hipFree(S8)); CK(hipFree(S4)); CK(hipFree(d_bad)); CK(hipFree(dh)); CK(hipFree(dt)); CK(hipFree(dref));
  }
  if (check) printf(fails ? "FAILED\n" : "all checks passed\n");
  return fails ? 1 : 0;
}
