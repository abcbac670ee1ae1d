// probe: throughput of sparse 64-bit integer atomic adds (device scope, no return) into a per-gene table, from every CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned long long* tab, size_t per_gene, int genes, int per_lane, unsigned seed) {
  unsigned x = seed ^ (blockIdx.x * 9781u + threadIdx.x * 6271u + 1u);
  for (int i = 0; i < per_lane; ++i) {
    x = x * 1664525u + 1013904223u;
    const unsigned g = (blockIdx.x + (x >> 28)) % genes;           // a WG touches a few genes
    const size_t e = (size_t)g * per_gene + ((x >> 4) % per_gene);
    __hip_atomic_fetch_add(tab + e, (unsigned long long)(x & 1023), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void kl(unsigned long long* tab, size_t per_gene, int genes, int per_lane, unsigned seed) {  // LDS version
  __shared__ unsigned long long t[6400];
  for (int i = threadIdx.x; i < 6400; i += blockDim.x) t[i] = 0;
  __syncthreads();
  unsigned x = seed ^ (blockIdx.x * 9781u + threadIdx.x * 6271u + 1u);
  for (int i = 0; i < per_lane; ++i) {
    x = x * 1664525u + 1013904223u;
    __hip_atomic_fetch_add(t + ((x >> 4) % 6400), (unsigned long long)(x & 1023), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 6400; i += blockDim.x) if (t[i]) tab[(size_t)(blockIdx.x % genes) * per_gene + i] = t[i];
}
int main() {
  const int genes = 512; const size_t per_gene = 9640;  // 77 KB
  unsigned long long* tab; hipMalloc(&tab, genes * per_gene * 8); hipMemset(tab, 0, genes * per_gene * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 2; ++mode)
  for (int per_lane : {4, 32, 256}) {
    const int blocks = 4096, threads = 256;
    if (mode == 0) k<<<blocks, threads>>>(tab, per_gene, genes, per_lane, 1); else kl<<<blocks, threads>>>(tab, per_gene, genes, per_lane, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) { if (mode == 0) k<<<blocks, threads>>>(tab, per_gene, genes, per_lane, r); else kl<<<blocks, threads>>>(tab, per_gene, genes, per_lane, r); }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = 5.0 * blocks * threads * per_lane;
    printf("%s per_lane %4d: %.3f ms per launch, %.2f G atomics/s\n", mode ? "LDS   " : "global", per_lane, ms / 5, n / ms / 1e6);
  }
  return 0;
}
