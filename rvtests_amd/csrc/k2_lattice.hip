// rvtests_amd — lattice-dosage sufficient-statistics kernels (suffstat_lat.hip.h), one instantiation per tile class; a
// translation unit of its own so that the engine's objects compile in parallel.
#include "suffstat_lat.hip.h"

namespace rvt {

// (ring depth, waves per SIMD) per tile class: the fastest of tools/k2lat_bench.hip on the widths of a batch (N = 500 000,
// isolated: 4.8 / 5.2 / 5.6 / 5.6 / 5.3 TB/s algorithmic); MT = 4 keeps its `hi` tiles in LDS and fits two waves per SIMD
// with the rolling refill, MT = 5 (354 registers) runs one.
void k2_launch_lat(int MT, dim3 grid, hipStream_t st, const GeneDesc* d_desc, NullTile nt, double den, long long N,
                   long long ld, int d) {
  const LatParam lp{den};
  switch (MT) {
    case 1: hipLaunchKernelGGL((gene_suffstat_lat<1, 2, 4>), grid, dim3(64), 0, st, d_desc, nt, lp, N, ld, d); break;
    case 2: hipLaunchKernelGGL((gene_suffstat_lat<2, 2, 2>), grid, dim3(64), 0, st, d_desc, nt, lp, N, ld, d); break;
    case 3: hipLaunchKernelGGL((gene_suffstat_lat<3, 1, 2>), grid, dim3(64), 0, st, d_desc, nt, lp, N, ld, d); break;
    case 4: hipLaunchKernelGGL((gene_suffstat_lat<4, 1, 2>), grid, dim3(64), 0, st, d_desc, nt, lp, N, ld, d); break;
    case 5: hipLaunchKernelGGL((gene_suffstat_lat<5, 2, 1>), grid, dim3(64), 0, st, d_desc, nt, lp, N, ld, d); break;
    default: break;
  }
}

}  // namespace rvt
