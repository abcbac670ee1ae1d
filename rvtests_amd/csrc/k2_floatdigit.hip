// rvtests_amd — the float-digit dosage sufficient-statistics kernel (suffstat_fdx.hip.h); a translation unit of its own so that
// the engine's objects compile in parallel.
#include "suffstat_fdx.hip.h"

namespace rvt {

// grid (wave-parts, genes), 8 waves per workgroup (4 loaders + 4 tile waves), one workgroup per CU.  MT = 0: the genes of every
// class up to kFdxEngineMT in one launch (the descriptors are sorted widest first).
void k2_launch_fdx(int MT, dim3 grid, hipStream_t st, const GeneDesc* d_desc, const NullTileF& nt, long long N, long long ld,
                   int d) {
  const dim3 block((kFdxNW + kFdxTW) * 64);
  switch (MT) {
    case 0: hipLaunchKernelGGL((gene_suffstat_fdx_any<kFdxEngineMT>), grid, block, 0, st, d_desc, nt, N, ld, d); break;
    case 1: hipLaunchKernelGGL((gene_suffstat_fdx<1>), grid, block, 0, st, d_desc, nt, N, ld, d); break;
    case 2: hipLaunchKernelGGL((gene_suffstat_fdx<2>), grid, block, 0, st, d_desc, nt, N, ld, d); break;
    case 3: hipLaunchKernelGGL((gene_suffstat_fdx<3>), grid, block, 0, st, d_desc, nt, N, ld, d); break;
    case 4: hipLaunchKernelGGL((gene_suffstat_fdx<4>), grid, block, 0, st, d_desc, nt, N, ld, d); break;
    default: break;
  }
}

}  // namespace rvt
