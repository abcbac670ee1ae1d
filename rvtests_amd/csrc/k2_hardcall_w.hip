// rvtests_amd — weighted (binary-trait) hard-call sufficient-statistics kernels (suffstat_hcw.hip.h), one
// instantiation per tile class; a translation unit of its own so that the engine's objects compile in parallel.
#include "suffstat_hcw.hip.h"

namespace rvt {

// (ring depth, waves per SIMD) per tile class: the fastest of tools/k2hcw_bench.hip (N = 200 000, isolated, with the
// value tests of round 3: 5.4 / 6.2 / 6.2 / 5.6 / 5.4 TB/s algorithmic); MT = 3 and MT = 5 keep two of their three pairs of
// int32 tiles in LDS (suffstat_hcw.hip.h) — MT = 3 runs two waves per SIMD that way, MT = 5 (rolling refill, depth 1)
// stops spilling
void k2_launch_hcw(int MT, dim3 grid, hipStream_t st, const GeneDesc* d_desc, NullTileW nt, long long N, long long ld,
                   int d) {
  switch (MT) {
    case 1: hipLaunchKernelGGL((gene_suffstat_hcw<1, 2, 3>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    case 2: hipLaunchKernelGGL((gene_suffstat_hcw<2, 2, 2>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    case 3: hipLaunchKernelGGL((gene_suffstat_hcw<3, 2, 2>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    case 4: hipLaunchKernelGGL((gene_suffstat_hcw<4, 2, 1>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    case 5: hipLaunchKernelGGL((gene_suffstat_hcw<5, 1, 1>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    default: break;
  }
}

}  // namespace rvt
