// rvtests_amd — translation unit of the unweighted (quantitative trait) sufficient-statistics kernels (see suffstat_kernels.hip.h);
// compiled in parallel with the other two objects of librvtests_amd.so.
#include "suffstat_kernels.hip.h"

namespace rvt {

void k2_launch_group_w0(int group, dim3 grid, hipStream_t st, const GeneDesc* d_desc, const int* list, int n_wparts,
                        NullDev nd, long long N, long long ld, int d) {
  const dim3 block(64);
  switch (group) {
    case 0: hipLaunchKernelGGL((gene_suffstat_mfma<0, false>), grid, block, 0, st, d_desc, list, n_wparts, nd, N, ld, d); break;
    case 1: hipLaunchKernelGGL((gene_suffstat_mfma<1, false>), grid, block, 0, st, d_desc, list, n_wparts, nd, N, ld, d); break;
    default: hipLaunchKernelGGL((gene_suffstat_mfma<2, false>), grid, block, 0, st, d_desc, list, n_wparts, nd, N, ld, d); break;
  }
}

void k2_launch_panel_w0(dim3 grid, hipStream_t st, const GeneDesc* d_desc, NullDev nd, long long N, long long ld,
                           int d) {
  hipLaunchKernelGGL((gene_suffstat_panel<false>), grid, dim3(64), 0, st, d_desc, nd, N, ld, d);
}

}  // namespace rvt
