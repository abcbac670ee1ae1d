// rvtests_amd — sufficient statistics from PLINK 2-bit rows (suffstat_hcp.hip.h), one instantiation per tile class; a
// translation unit of its own so that the engine's objects compile in parallel.
#include "suffstat_hcp.hip.h"

namespace rvt {

void k2_launch_hcp(int MT, dim3 grid, hipStream_t st, const GeneDesc* d_desc, NullTile nt, long long N, long long ld,
                   int d, const HcpPlanes* pl, bool score_only) {
  if (pl && score_only) {
    // single-variant score tests (rvt_score_bed_dev): T rows, the Gram diagonal and the column statistics from ONE kernel
    switch (MT) {
      case 1: hipLaunchKernelGGL((gene_tnull_hcp<1, true>), grid, dim3(64), 0, st, d_desc, *pl, N, ld); break;
      case 2: hipLaunchKernelGGL((gene_tnull_hcp<2, true>), grid, dim3(64), 0, st, d_desc, *pl, N, ld); break;
      default: break;
    }
    return;
  }
  if (pl) {
    // G'[X | rr] from the digit planes of the null tile (gene_tnull_hcp), everything else from the kernel without that product
    switch (MT) {
      case 1: hipLaunchKernelGGL((gene_suffstat_hcp<1, 4, false>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
      case 2: hipLaunchKernelGGL((gene_suffstat_hcp<2, 3, false>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
      case 3: hipLaunchKernelGGL((gene_suffstat_hcp<3, 2, false>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
      case 4: hipLaunchKernelGGL((gene_suffstat_hcp<4, 2, false>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
      case 5: hipLaunchKernelGGL((gene_suffstat_hcp<5, 1, false>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
      case 6: hipLaunchKernelGGL((gene_suffstat_hcp<6, 1, false>), grid, dim3(64), 0, st, d_desc, nt, N, ld, d); break;
      default: return;
    }
    switch (MT) {
      case 1: hipLaunchKernelGGL((gene_tnull_hcp<1, false>), grid, dim3(64), 0, st, d_desc, *pl, N, ld); break;
      case 2: hipLaunchKernelGGL((gene_tnull_hcp<2, false>), grid, dim3(64), 0, st, d_desc, *pl, N, ld); break;
      case 3: hipLaunchKernelGGL((gene_tnull_hcp<3, false>), grid, dim3(64), 0, st, d_desc, *pl, N, ld); break;
      case 4: hipLaunchKernelGGL((gene_tnull_hcp<4, false>), grid, dim3(64), 0, st, d_desc, *pl, N, ld); break;
      case 5: hipLaunchKernelGGL((gene_tnull_hcp<5, false>), grid, dim3(64), 0, st, d_desc, *pl, N, ld); break;
      case 6: hipLaunchKernelGGL((gene_tnull_hcp<6, false>), grid, dim3(64), 0, st, d_desc, *pl, N, ld); break;
      default: break;
    }
    return;
  }
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
