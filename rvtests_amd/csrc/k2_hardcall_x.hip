// rvtests_amd — the workgroup-cooperative weighted (binary-trait) hard-call sufficient-statistics kernels
// (suffstat_hcx.hip.h), one instantiation per tile class; a translation unit of its own so that the engine's objects compile
// in parallel.
#include "suffstat_hcx.hip.h"

namespace rvt {

// grid (wave-parts, genes of the class — MT = 0: genes of every class, one launch), 8 waves per workgroup (4 loaders + 4 tile waves), one workgroup per CU
void k2_launch_hcx(int MT, dim3 grid, hipStream_t st, const GeneDesc* d_desc, const NullTileX& nt, long long N, long long ld,
                   int d) {
  const dim3 block(2 * kHcxNW * 64);
  switch (MT) {
    case 0: hipLaunchKernelGGL((gene_suffstat_hcx_any<kHcxMaxMT>), grid, block, 0, st, d_desc, nt, N, ld, d); break;  // (all classes)
    case 1: hipLaunchKernelGGL((gene_suffstat_hcx<1>), grid, block, 0, st, d_desc, nt, N, ld, d); break;
    case 2: hipLaunchKernelGGL((gene_suffstat_hcx<2>), grid, block, 0, st, d_desc, nt, N, ld, d); break;
    case 3: hipLaunchKernelGGL((gene_suffstat_hcx<3>), grid, block, 0, st, d_desc, nt, N, ld, d); break;
    case 4: hipLaunchKernelGGL((gene_suffstat_hcx<4>), grid, block, 0, st, d_desc, nt, N, ld, d); break;
    case 5: hipLaunchKernelGGL((gene_suffstat_hcx<5>), grid, block, 0, st, d_desc, nt, N, ld, d); break;
    default: break;
  }
}

}  // namespace rvt
