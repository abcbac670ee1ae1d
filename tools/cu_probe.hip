// Probe: which (XCC, SE, CU) a CU-masked stream really runs on.  usage: cu_probe <hex words, least significant first>...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void probe(unsigned* out) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // spin a little so that blocks spread over every enabled CU
  long long t0 = clock64();
  while (clock64() - t0 < 200000) {}
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xff00);
}
int main(int argc, char** argv) {
  std::vector<unsigned> mask;
  for (int i = 1; i < argc; ++i) mask.push_back((unsigned)strtoul(argv[i], nullptr, 16));
  hipStream_t st;
  if (mask.empty()) hipStreamCreate(&st);
  else if (hipExtStreamCreateWithCUMask(&st, mask.size(), mask.data()) != hipSuccess) { printf("mask create failed\n"); return 1; }
  const int nb = 8192;
  unsigned* d; hipMalloc(&d, nb * 4);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(64), 0, st, d);
  hipStreamSynchronize(st);
  std::vector<unsigned> h(nb); hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, int> hist;
  for (unsigned v : h) hist[v]++;
  std::map<int, int> per_xcc;
  printf("distinct (xcc,se,sh,cu): %zu\n", hist.size());
  for (auto& kv : hist) {
    unsigned xcc = kv.first >> 16, hw = kv.first & 0xffff;
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcc[xcc]++;
    if (hist.size() <= 80) printf("  xcc %u se %u sh %u cu %u : %d blocks\n", xcc, se, sh, cu, kv.second);
  }
  for (auto& kv : per_xcc) printf("xcc %d: %d CUs\n", kv.first, kv.second);
  return 0;
}
