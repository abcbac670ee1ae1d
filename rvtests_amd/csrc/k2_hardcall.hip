// rvtests_amd — hard-call sufficient-statistics kernels (suffstat_hc.hip.h), one instantiation per tile class; a
// translation unit of its own so that the engine's objects compile in parallel.
#include "suffstat_hc.hip.h"

namespace rvt {

// (ring depth, waves per SIMD) per tile class: every class streams at the HBM rate with two steps in flight
// (tools/k2hc_bench.hip: 5.9 - 6.9 TB/s isolated for every (depth, waves) tried), so each takes the smallest register
// budget — the fewer registers, the more easily its waves share a SIMD with the latency-bound kernels of other batches.
void k2_launch_hc(int MT, dim3 grid, hipStream_t st, const GeneDesc* d_desc, NullTile nt, long long N, long long ld,
                  int d) {
  switch (MT) {
    case 1: hipLaunchKernelGGL((gene_suffstat_hc<1, 2, 4, false>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    case 2: hipLaunchKernelGGL((gene_suffstat_hc<2, 2, 3, false>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    case 3: hipLaunchKernelGGL((gene_suffstat_hc<3, 2, 2, false>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    case 4: hipLaunchKernelGGL((gene_suffstat_hc<4, 2, 2, false>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    // MT = 5: rolling refill (depth 1) fits 256 registers, so two waves share a SIMD — or one shares it with a
    // 256-register p-value wave, which a 311-register wave cannot (measured live: 3.5 -> TB/s)
    case 5: hipLaunchKernelGGL((gene_suffstat_hc<5, 1, 1, false>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    case 6: hipLaunchKernelGGL((gene_suffstat_hc<6, 1, 1, false>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    default: break;
  }
}

void k2_launch_classify(dim3 grid, hipStream_t st, const double* G, long long N, long long ld, int M, int* flag) {
  hipLaunchKernelGGL(block_classify_kernel<0>, grid, dim3(256), 0, st, G, N, ld, M, flag);
}

}  // namespace rvt
