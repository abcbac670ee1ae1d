// rvtests_amd — sufficient statistics from PLINK 2-bit rows (suffstat_hcp.hip.h), one instantiation per tile class; a
// translation unit of its own so that the engine's objects compile in parallel.
#include "suffstat_hcp.hip.h"

namespace rvt {

void k2_launch_hcp(int MT, dim3 grid, hipStream_t st, const GeneDesc* d_desc, NullTile nt, long long N, long long ld,
                   int d) {
  switch (MT) {
    case 1: hipLaunchKernelGGL((gene_suffstat_hcp<1, 4>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    case 2: hipLaunchKernelGGL((gene_suffstat_hcp<2, 3>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    case 3: hipLaunchKernelGGL((gene_suffstat_hcp<3, 2>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    case 4: hipLaunchKernelGGL((gene_suffstat_hcp<4, 2>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    case 5: hipLaunchKernelGGL((gene_suffstat_hcp<5, 1>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    case 6: hipLaunchKernelGGL((gene_suffstat_hcp<6, 1>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
    default: break;
  }
}

}  // namespace rvt
