// rvtests_amd — plain-old-data records exchanged between the kernels of the per-gene pipeline.
#pragma once
#include <stdint.h>
#include "../../include/rvtests_amd.h"

namespace rvt {

constexpr int kMaxLambda = 0;  // lambdas live in a side buffer, see GeneStats::lambda_off

// Null-model constants shared by every gene (built once by rvt_set_null).
// Follows what SkatTest/SkatOTest/CMCTest cache between genes (src/Model.h:2672-2699).
struct NullConsts {
  int64_t N;
  int64_t ld;      // padded sample count of device arrays (multiple of 16)
  int d;           // columns of X incl. intercept
  int binary;      // 0 quantitative, 1 binary
  double sigma2;   // RSS/N (quantitative), 1 otherwise        LinearRegression.cpp:60
  double rss;      // sum res^2
  double rsum;     // sum res
  double C[RVT_MAX_COV * RVT_MAX_COV];     // X'VX (binary) or X'X (quantitative), row-major d x d
  double Cinv[RVT_MAX_COV * RVT_MAX_COV];  // inverse of C
};

// Output of the statistics kernel (gene_stats_kernel), input of the p-value kernel.
struct GeneStats {
  int status;        // RVT_ST_* bits
  int n_variants;    // M as submitted
  int n_poly;        // after flip + monomorphic removal
  int flip_count;
  // SKAT
  double skat_Q;
  int skat_nlambda;  // kept eigenvalues (> 1e-30, descending)        Skat.cpp:87-98
  int skat_lambda_off;
  // SKAT-O
  int skato_single;  // 1: single-variant shortcut (FitSKAT)       SkatO.cpp:60-99,118-120
  int skato_ok;      // stage A reached the SKAT-O part (see skato_fit_ok for the eigen checks)
  int eig_ok[13];    // per eigenproblem: 1 if getEigen found a positive eigenvalue
  double Qs[11];
  double mom_mu[11], mom_var[11], mom_df[11];
  double tau[11];
  double muQ, varQ, varZeta, df;
  int zimz_nlambda;
  int zimz_lambda_off;
  double zimz_lambda_sum;
  // burden
  int cmc_nonref;
  int cmc_ok, zeg_ok;
  double cmc_U, cmc_V, cmc_stat;
  double zeg_U, zeg_V, zeg_stat;
#ifdef RVT_PROF_K4
  double as_ticks[10];  // profiling build: clock at the phase boundaries of gene_assemble (tools/pv_prof.py)
#endif
};

}  // namespace rvt
