/* rvtests_amd — C ABI of the MI355X kernel/burden association engine.
 *
 * This is the drop-in boundary for the rvtests hot path: the entry points below are what GPU-backed
 * `ModelFitter` subclasses (SkatTest, SkatOTest, CMCTest, ZegginiTest, FamSkatTest, MetaCovTest, MetaScoreTest — see
 * rvtests_amd/csrc/host/ModelFitterGpu.h and INTEGRATION.md) call instead of
 *   Skat::Fit                         /root/reference/regression/Skat.h:26-31   (Skat.cpp:29-105)
 *   SkatO::Fit                        regression/SkatO.h:28-35                  (SkatO.cpp:101-281,500-519)
 *   cmcCollapse / zegginiCollapse     src/Model.cpp:73-89,115-130
 *   LinearRegressionScoreTest::TestCovariate(Matrix,Vector,Matrix)      regression/LinearRegressionScoreTest.cpp:173-263
 *   LogisticRegressionScoreTest::TestCovariate(Matrix,Vector,Matrix)    regression/LogisticRegressionScoreTest.cpp:220-302
 *   DataConsolidator::getFlippedToMinorPolymorphicGenotype               src/DataConsolidator.h:128-132
 *   Skat::GetQFromNewResidual + Permutation (skat[nPerm>0])               regression/Skat.cpp:107-116, src/Permutation.h:69-98
 *   LinearRegression::FitLinearModel / LogisticRegression::FitLogisticModel   regression/LinearRegression.cpp:20-69,
 *                                                                             regression/LogisticRegression.cpp:279-336
 *   FamSkat::FitNullModel / TestCovariate, FastLMM::FitNullModel / FastGetAF  regression/FamSkat.cpp:34-138,
 *                                                                             regression/FastLMM.cpp:28-142,402-443
 *   MetaCovTest (MetaCovUnrelatedQtl / UnrelatedBinary / FamQtl)          src/Model.cpp:437-1004
 *   MetaScoreTest (MetaUnrelatedQtl / MetaUnrelatedBinary / MetaFamQtl)   src/Model.h:3155-3784
 *   AnalyticVT (UNRELATED) over MultivariateVT::compute                  src/Model.h:2105-2259, regression/MultivariateVT.cpp:22-144
 *   KBACTest over KbacTest::calcKbacP                                     src/Model.h:2891-3045, regression/kbac.cpp:16-391
 *   VCFGenotypeExtractor::extractMultipleGenotype (GT text)               src/VCFGenotypeExtractor.cpp:29-140,397-439
 *   BGenFile::parseLayout1 / parseLayout2 (probability blocks),           libBgen/BGenFile.cpp:205-238,321-392,
 *   BGenGenotypeExtractor::getGenotype                                    src/BGenGenotypeExtractor.cpp:413-478
 *   KinshipHolder::decompose                                              base/KinshipHolder.cpp:270-290
 * Plain C, plain pointers and sizes; no C++ or torch types cross it.  All functions return 0 on
 * success and a negative RVT_E_* code on failure; rvt_last_error() gives the text.  One calling
 * thread per context; one context per GPU (one process per GPU).
 *
 * Data layout.  Matrices are column-major doubles exactly as the reference's `Matrix`
 * (base/MathMatrix.h:33-41,107-110: data[i + j*rows]).  On the device every sample-indexed array is
 * padded to a leading dimension ld = rvt_padded_ld(N) (next multiple of 16, i.e. whole 128-byte
 * lines) and the pad rows are zero.
 */
#ifndef RVTESTS_AMD_H_
#define RVTESTS_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVT_MAX_COV 16 /* max columns of X (intercept included) */

/* error codes */
#define RVT_OK 0
#define RVT_E_INVALID (-1)   /* bad argument */
#define RVT_E_NO_DEVICE (-2) /* no usable HIP device: the engine has NO CPU fallback */
#define RVT_E_HIP (-3)       /* a HIP call failed */
#define RVT_E_STATE (-4)     /* call sequence error (e.g. no null model set) */
#define RVT_E_TOO_LARGE (-5) /* gene wider than RVT_MAX_VARIANTS */

#define RVT_MAX_VARIANTS 1024
#ifndef RVT_MAX_INFLIGHT
#define RVT_MAX_INFLIGHT 8 /* batches that rvt_run_blocks_async keeps in flight */
#endif

/* test selection bitmask (which ModelFitter::fit bodies to run) */
#define RVT_TEST_SKAT 1u    /* --kernel skat   : SkatTest    src/Model.h:2612-2772 */
#define RVT_TEST_SKATO 2u   /* --kernel skato  : SkatOTest   src/Model.h:2774-2889 */
#define RVT_TEST_CMC 4u     /* --burden cmc    : CMCTest     src/Model.h:807-907   */
#define RVT_TEST_ZEGGINI 8u /* --burden zeggini: ZegginiTest src/Model.h:1170-1242 */
#define RVT_TEST_ALL 15u /* the four gene-level tests of the headline workload */
#define RVT_TEST_FAMSKAT 16u /* --kernel famSkat: FamSkatTest src/Model.h:3048-3145 (rvt_run_fam_tests only) */
#define RVT_TEST_FAMCMC 32u     /* --burden famcmc    : FamCMC     src/Model.h:2261-2376 (rvt_run_fam_tests only) */
#define RVT_TEST_FAMZEGGINI 64u /* --burden famzeggini: FamZeggini src/Model.h:2378-2492 (rvt_run_fam_tests only) */
#define RVT_TEST_ANALYTICVT 128u /* --vt analytic : AnalyticVT (UNRELATED) src/Model.h:2105-2259, quantitative traits only */

/* Evaluate the coefficient sums of Davies' qf() term by term, in the reference's order (one atan and one log per
 * coefficient and term, regression/qfc.c:143-152,192-205,250-262), instead of the product form the engine uses by
 * default (rvtests_amd/csrc/rvt_davies.h: same sums to ~1e-15 absolute, ~8x fewer instructions).  For verification. */
#define RVT_TEST_EXACT_DAVIES 0x80000000u

/* trait type of the null model */
#define RVT_TRAIT_QUANTITATIVE 0
#define RVT_TRAIT_BINARY 1

/* per-gene status bits */
#define RVT_ST_NO_POLY 1u      /* no polymorphic variant left: every test prints NA (fit returns -1) */
#define RVT_ST_SKATO_EIGEN 2u  /* SkatO::Fit returned -1 (no positive eigenvalue)  SkatO.cpp:170-175 */
#define RVT_ST_CMC_FAIL 4u
#define RVT_ST_ZEG_FAIL 8u
#define RVT_ST_INPUT_ERROR 16u /* streaming VCF-text / BGEN submission: a record of THIS gene was malformed (wrong column
                                  count, inexact dosage, truncated block); every test is NA, rvt_last_error names the
                                  record.  With af_out != NULL the submit call itself returns RVT_E_INVALID instead. */

typedef struct rvt_ctx rvt_ctx;

/* model parameters, the tags ModelParser hands to the models (src/ModelManager.cpp:168-198) */
typedef struct rvt_params {
  double skat_beta1, skat_beta2;   /* skat[beta1=1:beta2=25]  */
  double skato_beta1, skato_beta2; /* skato[beta1=1:beta2=25] */
  int skat_nperm;                  /* skat[nPerm=...]; 0 = analytic p-value only */
  double skat_alpha;               /* skat[alpha=0.05] */
} rvt_params;

/* one record per gene, in submission order — the numbers the models' writeOutput() print */
typedef struct rvt_gene_result {
  int64_t gene_id;
  uint32_t status;  /* RVT_ST_* */
  int n_variants;   /* columns submitted */
  int n_poly;       /* columns after flip-to-minor + monomorphic removal ("NumPolyVar") */
  /* SKAT: "Q\tPvalue"  (src/Model.h:2722-2749) */
  int skat_ok;
  double skat_Q, skat_p;
  int skat_nlambda;
  /* SKAT-O: "Q\trho\tPvalue"  (src/Model.h:2862-2875) */
  int skato_ok;
  double skato_Q, skato_rho, skato_p;
  int skato_qags_status; /* GSL error number of the Davies-integrand integration (0 = converged) */
  int skato_qags_neval;  /* integrand evaluations */
  /* CMC: "NonRefSite\tPvalue"  (src/Model.h:860-888) */
  int cmc_ok, cmc_nonref;
  double cmc_U, cmc_V, cmc_stat, cmc_p;
  /* Zeggini: "Pvalue"  (src/Model.h:1217-1231) */
  int zeg_ok;
  double zeg_U, zeg_V, zeg_stat, zeg_p;
  /* diagnostics */
  double davies_terms; /* integrand terms evaluated by all Davies calls of this gene */
  /* SKAT permutation columns "NumPerm ActualPerm Stat NumGreater NumEqual PermPvalue" (src/Permutation.h:40-141);
   * Stat = skat_Q.  Filled when rvt_params.skat_nperm > 0 (synchronous entry points only). */
  int perm_ok;
  int perm_num_perm, perm_actual_perm, perm_num_greater, perm_num_equal;
  double perm_pvalue;
  /* FamSKAT: "Q\tPvalue"  (src/Model.h:3121-3132); famskat_p may be -1 (Davies fault, no Liu fallback there) */
  int famskat_ok;
  double famskat_Q, famskat_p;
  /* FamCMC "NumSite AF U V Effect Pvalue" / FamZeggini "NumSite MeanBurden U V Effect Pvalue" (NumSite = n_poly,
   * Effect = U / V); src/Model.h:2344-2357, 2461-2473 */
  int famcmc_ok;
  double famcmc_af, famcmc_U, famcmc_V, famcmc_p;
  int famzeg_ok;
  double famzeg_af, famzeg_U, famzeg_V, famzeg_p;
  /* AnalyticVT "MinMAF MaxMAF OptimMAF OptimNumVar U V Stat Pvalue" (src/Model.h:2233-2245 over
   * MultivariateVT::compute, regression/MultivariateVT.cpp:22-144).  vt_p is 1 - P(|Z_i| < Stat for every threshold)
   * for the correlation of the nested threshold statistics: the reference evaluates that integral with a RANDOMISED
   * rule at absolute accuracy 1e-3 (regression/libMvtnorm); here it is evaluated deterministically to ~1e-5
   * (rvtests_amd/csrc/rvt_mvn.h); vt_p_error is the estimated absolute error.  vt_ncutoff = number of distinct MAF
   * thresholds (the dimension of the integral). */
  int vt_ok, vt_optnum, vt_ncutoff;
  double vt_minmaf, vt_maxmaf, vt_optmaf, vt_U, vt_V, vt_stat, vt_p, vt_p_error;
} rvt_gene_result;

/* FastLMM null model of the related-sample tests, as FamSkat::FitNullModel consumes it */
typedef struct rvt_fam_null {
  double delta;             /* sigma2_e / sigma2_g, FastLMM::GetDelta()              */
  double sigma2_g;          /* FastLMM::GetSigmaG2()                                 */
  double beta[RVT_MAX_COV]; /* FastLMM::GetBeta()                                    */
  int max_index;            /* best point of the 101-point delta grid                */
  int brent_evals;          /* goal-function evaluations of the Brent refinement     */
} rvt_fam_null;

/* accumulated device time per kernel family, measured with HIP events on the engine's stream */
typedef struct rvt_timing {
  double ms_suffstat;   /* gene_suffstat_hc + gene_suffstat_mfma launches      */
  double ms_burden;     /* burden_collapse                                     */
  double ms_stats;      /* gene_stats (flip algebra, eigen, moments)           */
  double ms_pvalue;     /* gene_pvalue (Davies / Liu / QAGS)                   */
  int64_t n_suffstat_launches, n_burden_launches, n_stats_launches, n_pvalue_launches;
  int64_t genes;        /* genes processed while profiling was on              */
  double alg_bytes;     /* algorithmic bytes of those genes: 8*N*M + 8*N*(d+2) each */
  double alg_flops;     /* algorithmic flops: 2*N*M*(M+d+1) each                                 */
  int64_t genes_hard_call; /* of `genes`: those that took the hard-call (int8 matrix core) kernel        */
  double ms_suffstat_hc;   /* of ms_suffstat: the gene_suffstat_hc launches                              */
  int64_t n_suffstat_hc_launches;
  double alg_bytes_hc;     /* of alg_bytes: the hard-call genes                                           */
  int64_t genes_handed_back; /* of genes_hard_call: those whose block held something else (dosages) and that were
                                computed again by the fp64 kernel                                              */
} rvt_timing;

/* ---- lifetime ------------------------------------------------------------------------------- */
/* Create a context on HIP device `device_id`.  Fails with RVT_E_NO_DEVICE when there is no GPU. */
int rvt_init(rvt_ctx** ctx, int device_id);
void rvt_destroy(rvt_ctx* ctx);
const char* rvt_last_error(const rvt_ctx* ctx);
const char* rvt_version(void);
/* padded leading dimension for N samples (multiple of 16) */
int64_t rvt_padded_ld(int64_t N);

/* ---- null model ------------------------------------------------------------------------------
 * What SkatTest::fit caches after LinearRegression::FitLinearModel / LogisticRegression::
 * FitLogisticModel (src/Model.h:2672-2699): X is N x d column-major INCLUDING the intercept column
 * (copyCovariateAndIntercept, src/ModelUtil.h:102-130), res = y - yhat, v = per-sample variance
 * (sigma2 for a quantitative trait, p(1-p) for a binary one).  Host pointers; copied. */
int rvt_set_null(rvt_ctx* ctx, int trait, int64_t N, int d, const double* X, const double* res,
                 const double* v, double sigma2);

/* ---- device-resident genotype blocks ------------------------------------------------------------
 * A block is the imputed, UNFLIPPED genotype matrix of one gene (what dc->getGenotype() holds when
 * fit() is called), N x M doubles, column-major with leading dimension rvt_padded_ld(N), pad = 0. */
int rvt_block_alloc(rvt_ctx* ctx, int M, double** dG_out);
int rvt_block_free(rvt_ctx* ctx, double* dG);
/* copy a host N x M column-major matrix (leading dimension N) into a block */
int rvt_block_upload(rvt_ctx* ctx, double* dG, int M, const double* G_host);
/* Page-locking the caller's hand-off buffer.  The reference fills ONE genotype buffer per gene and reuses it for the next
 * (src/Main.cpp:1086,1225), so every submission has to have read its input when it returns: by default the engine copies
 * pageable memory into a pinned ring with a few CPU threads (host_stage.h; ~35 GB/s) or lets the runtime do it.  An
 * adapter that owns the buffer can page-lock it ONCE — rvt_host_register(ctx, ptr, bytes) — and every later rvt_submit_* /
 * rvt_block_upload whose source lies inside a registered range is a DMA straight out of it at the rate of the link, with
 * no CPU copy (fp64 blocks that hold hard calls and imputed means are still packed to 2-bit rows by the staging threads first:
 * reading 200 MB at memory speed and sending 6 MB beats sending 200 MB at link speed); the call still returns only when the
 * range has been read.  rvt_host_unregister before the buffer is freed
 * (rvt_destroy unregisters what is left).  Registration changes no result. */
int rvt_host_register(rvt_ctx* ctx, const void* ptr, size_t bytes);
int rvt_host_unregister(rvt_ctx* ctx, const void* ptr);
/* Bind the calling thread — and every thread created after it, the engine's staging pools included — to the CPUs of the NUMA node
 * the device hangs on (what `numactl --cpunodebind` does from outside).  The hand-offs that pack on the host (rvt_submit_gene's
 * fp64 blocks, rvt_submit_gene_i8, the columns of rvt_block_upload_columns) read the caller's buffers with a few threads and
 * write a pinned ring on the device's node: with everything on that node the MetaCov adapter runs at 24 k instead of 18 k sites/s
 * on a two-socket host, the fp64 gene hand-off 15 % faster.  Needs no context (call it first in main()); returns the node, or -1
 * when it is unknown or the mask cannot be set (nothing changed then).  Changes no result. */
int rvt_pin_to_device_node(int device_id);
/* What the host side of the hand-off has to work with on THIS machine, measured in place (about a third of a second): a
 * figure from host memory that is far below another box's is explained by these numbers or by nothing the engine controls.
 * Rates in GB/s on a 64 MB buffer, best of three. */
typedef struct rvt_host_diag {
  int hardware_threads;        /* std::thread::hardware_concurrency()                                            */
  int affinity_cpus;           /* CPUs in this process's affinity mask (sched_getaffinity)                       */
  int copy_threads;            /* threads that copy pageable memory into the pinned ring (RVT_COPY_THREADS)      */
  int pack_threads;            /* threads of the fp64 -> 2-bit packing pass (RVT_PACK_THREADS)                   */
  int thp;                     /* transparent huge pages: 0 never, 1 madvise, 2 always, -1 unknown               */
  int gpu_numa_node;           /* NUMA node of the device (sysfs), -1 unknown                                    */
  int buffer_numa_node;        /* NUMA node the test buffer's pages landed on (first touch by the caller), -1    */
  int pinned_numa_node;        /* NUMA node of the pinned staging memory (hipHostMalloc), -1                     */
  double memcpy_one_thread;    /* pageable -> pageable memcpy, the calling thread                                */
  double stage_pool;           /* pageable -> pinned with the copy pool (what a staged hand-off does per gene)   */
  double h2d_pinned;           /* pinned -> device DMA                                                           */
  double d2h_pinned;           /* device -> pinned DMA                                                           */
  double loadavg1;             /* 1-minute load average of the machine                                           */
} rvt_host_diag;
int rvt_host_diagnose(rvt_ctx* ctx, rvt_host_diag* out);
/* Hard calls.  Where the entries of a block are exactly 0.0, 1.0 or 2.0 and the null model is unweighted (quantitative
 * trait), G'G is an integer matrix and the engine computes it on the int8 matrix cores instead of the fp64 ones, with
 * the burden collapse in the same pass (rvtests_amd/csrc/suffstat_hc.hip.h): same numbers (the integer part exactly,
 * G'X and G'r by the same fp64 products), and the kernel is bound by HBM alone.  A column that imputeGenotypeToMean
 * (src/DataConsolidator.cpp:217-245) has filled — hard calls plus ONE other value in the places of the missing calls —
 * stays on that kernel: the other value is carried as a 0/1 mask, three integer matrices and one multiplication per
 * entry of G'G.  Under a binary trait (weights v = p (1 - p)) the weighted Gram matrix takes the int8 cores as well, with
 * v split once per null model into six 7-bit digit planes (M <= 80, weights in [0, 0.49], agreement with the fp64 kernel
 * ~1e-12 relative).  Round 4 (suffstat_hcx.hip.h, gene tests with d <= 6 covariate columns): a workgroup of eight waves per
 * wave-part — four stream and pack, four multiply —, the null-model tile [vX | res | v] on the int8 cores too (six digit
 * planes per column, a power-of-two scale per column; a column whose largest entry exceeds 256 x its root mean square
 * keeps the model on the one-wave kernel suffstat_hcw.hip.h), every statistic an exact integer of the quantised inputs,
 * and mean-imputed columns stay on the kernel (sparse integer tables of the masked entries).  RVT_HCX=0 selects the
 * one-wave kernel, which hands a gene with an imputed column to the fp64 kernel.
 * NOTHING is remembered about the content of a block.  The integer kernels test every value they load; a block that
 * holds anything else (dosages) is handed back and computed by the fp64 kernel in the same call — the records are the
 * same either way.  Which kernel a block starts on is a prediction: what the engine's own decoders wrote; for the
 * caller's doubles (rvt_submit_gene, rvt_submit_gene_raw, rvt_run_blocks*): hard calls, unless the caller has said
 * otherwise with rvt_set_content_hint(ctx, 0) — what an adapter does when the analysis reads dosages (`--dosage`,
 * src/Main.cpp FLAG_dosageTag; BGEN input) and the test blocks would only be handed back.  A wrong hint costs time, never
 * correctness.  hint: -1 unknown (default), 0 dosages, 1 hard calls / mean-imputed hard calls.
 * rvt_block_classify is a query for tools and tests: one streaming pass, 1 when every entry is 0.0 / 1.0 / 2.0. */
int rvt_block_classify(rvt_ctx* ctx, const double* dG, int M, int* is_hard_call);
int rvt_set_content_hint(rvt_ctx* ctx, int hint);
/* Which sufficient-statistics kernel the hard-call genes of the installed null model take (for tools: bench.py names the
 * kernel its roofline line is about): 0 none (no null model / hard-call path off), 1 gene_suffstat_hc (quantitative trait),
 * 2 gene_suffstat_hcw (binary trait, one wave per part), 3 gene_suffstat_hcx (binary trait, cooperative workgroups). */
int rvt_hardcall_kernel(const rvt_ctx* ctx);
/* Dosages on a decimal lattice.  `rvtest --dosage DS` (src/Main.cpp FLAG_dosageTag; VCFGenotypeExtractor) hands fit() the
 * doubles strtod made of a VCF field printed with a fixed number of decimals — imputation servers write three — i.e. the
 * doubles nearest to K / denominator, K an integer, denominator = 10^decimals.  When the adapter states the denominator
 * (1 .. 2048; 0 = not stated, the default) such blocks — the caller's doubles under rvt_set_content_hint(ctx, 0), and
 * what rvt_submit_gene_vcf_dosage decodes — take gene_suffstat_lat (rvtests_amd/csrc/suffstat_lat.hip.h) under a
 * quantitative trait: K = rint(g denominator) is split into two 7-bit digits and G'G = K'K / denominator^2 is formed on
 * the int8 matrix cores, exactly, with the burden collapse in the same pass; the kernel is bound by HBM where the fp64
 * kernel is bound by the fp64 matrix pipe.  Every value is tested (|g denominator - K| <= K 2^-53 — i.e. g is the double nearest to K / denominator —, 0 <= g <= 2): a block that
 * holds anything else (BGEN's float probabilities, mean-imputed entries, a different number of decimals) is handed back
 * and computed by the fp64 kernel in the same call.  A wrong statement costs time, never correctness. */
int rvt_set_dosage_lattice(rvt_ctx* ctx, int denominator);
/* Float-precision dosages.  BGEN input reaches fit() as dosages formed from FLOAT probabilities
 * (src/BGenGenotypeExtractor.cpp:413-478: prob[] is std::vector<float>, dosage = p1 + 2 p2): integer multiples of 2^-31 for
 * 8-bit files, and any float-precision dosage that is 0 or >= 2^-14 is a multiple of 2^-37.  Under a quantitative trait
 * such blocks — what rvt_submit_gene_bgen decodes, and the caller's doubles under rvt_set_content_hint(ctx, 0) — take
 * gene_suffstat_fdx WHEN rvt_set_dosage_float(ctx, 1) ASKS FOR IT (no lattice denominator stated; round 5: opt-in — the
 * kernel gives G'G as an exact integer but runs no faster than the fp64 kernel, so nothing starts on it unasked)
 * (rvtests_amd/csrc/suffstat_fdx.hip.h) for M <= 64: K = g 2^37 is read off the double, split into five balanced base-256
 * digits and K'K is formed exactly on the int8 matrix cores.  Every value is tested (g 2^37 an integer, 0 <= g <= 2): a
 * block that holds anything else is handed back and computed by the fp64 kernel in the same call.  A wrong statement costs
 * time, never correctness.  RVT_FDX=0 in the environment keeps such blocks on the fp64 kernel. */
int rvt_set_dosage_float(rvt_ctx* ctx, int on);
/* experiments / tests: on = 0 keeps every gene on the fp64 kernel (as the environment variable RVT_HARDCALL=0 does for
 * the whole process), on = 1 restores the default */
int rvt_set_hardcall(rvt_ctx* ctx, int on);

/* Run the selected tests on n_genes blocks that are already in HBM.  dG[g] are DEVICE pointers
 * (rvt_block_alloc, or any 128-byte aligned device allocation with the layout above, e.g. a torch
 * tensor); M[g] the column counts; af is the HOST concatenation of the per-column allele
 * frequencies GenotypeCounter::getAF() reports (src/GenotypeCounter.h:46-51), sum(M) entries.
 * Blocks until the results are in out[0..n_genes); lists longer than 256 genes are processed as pipelined batches of
 * 256.  */
int rvt_run_blocks(rvt_ctx* ctx, int n_genes, const double* const* dG, const int* M, const double* af,
                   const int64_t* gene_ids, uint32_t tests, const rvt_params* params, rvt_gene_result* out);
/* Same, but only enqueue; up to RVT_MAX_INFLIGHT batches may be in flight (each on its own HIP stream, so that
 * the latency-bound tail of one overlaps the bandwidth-bound head of the next).  A batch's records are valid after
 * rvt_sync() (everything) or rvt_wait_oldest() (the batch launched first); launching one more first
 * finishes the oldest one.  `out` must stay alive until then. */
int rvt_run_blocks_async(rvt_ctx* ctx, int n_genes, const double* const* dG, const int* M, const double* af,
                         const int64_t* gene_ids, uint32_t tests, const rvt_params* params,
                         rvt_gene_result* out);
int rvt_sync(rvt_ctx* ctx);
int rvt_wait_oldest(rvt_ctx* ctx);
/* Allocate, for all RVT_MAX_INFLIGHT pipeline slots at once, the device workspace and pinned staging a batch of
 * n_genes genes with column counts M[] needs (otherwise each slot grows on its first use: a hipMalloc /
 * hipHostMalloc of some hundred milliseconds inside the first batches).  Needs the null model (defines N, d). */
int rvt_reserve(rvt_ctx* ctx, int n_genes, const int* M);

/* ---- streaming interface used by the ModelFitter adapters ------------------------------------------
 * rvt_submit_gene copies G (host, N x M column-major, imputed, unflipped) before returning — the
 * caller's buffer is overwritten by the next gene (src/Main.cpp:1086,1225).  Results come back in
 * submission order from rvt_collect (which flushes pending genes). */
int rvt_submit_gene(rvt_ctx* ctx, int64_t gene_id, int M, const double* G, const double* af, uint32_t tests,
                    const rvt_params* params);
int rvt_collect(rvt_ctx* ctx, rvt_gene_result* out, int cap, int* n_out);
/* Non-blocking companion: the records of the longest FINISHED prefix of the submitted genes (at most cap), without
 * launching incomplete groups and without waiting for anything — the pipeline keeps running.  *n_out may be 0.  A
 * caller that streams genes (the ModelFitter adapters) polls this while it submits and calls rvt_collect only at the
 * end, so the device never drains between windows. */
int rvt_collect_ready(rvt_ctx* ctx, rvt_gene_result* out, int cap, int* n_out);

/* ---- MetaCov: score-covariance band (`--meta cov`) ------------------------------------------------------
 * Replaces the arithmetic of MetaCovTest::fitWithGivenGenotype / printCovariance / computeScaledXX
 * (src/Model.cpp:844-1004, src/Model.h:3993-4005) with MetaCovUnrelatedQtl / MetaCovUnrelatedBinary
 * (src/Model.cpp:506-593, 694-778) as the model — i.e. unrelated samples; rvt_cov_block_fam below is the family
 * (FastLMM) counterpart.  `dG` is a device block (rvt_block_alloc) whose V <= RVT_MAX_VARIANTS columns are the imputed genotype
 * vectors of V consecutive single-variant fit() calls (what assignGenotype copies into its ring, Model.cpp:936-942).
 * Outputs (host):
 *   cov[h + j*V], j >= h : covXX(h,j) - covXZ_h' covZZInv covXZ_j, the value printCovariance prints for head h and
 *                          marker j BEFORE its 1/N scaling (lower triangle untouched)
 *   xz[h*d + k]          : covXZ of variant h (printed after ':' for binary traits / gwama)
 *   zz[a*d + b]          : covZZ (may be NULL)
 *   polymorphic[h]       : 0 when isMonomorphicMarker (src/DataConsolidator.cpp:94-116) would have skipped h
 * The caller applies the window rule (src/Model.h:3956-3990) and the text formatting; the null model is the one
 * installed by rvt_set_null (trait, X with intercept, sigma2 / v).  Synchronous. */
int rvt_cov_block(rvt_ctx* ctx, const double* dG, int V, double* cov, double* xz, double* zz, int* polymorphic);
/* The same numbers for windows wider than one block (unrelated samples): heads = columns [col0, col0+H) of `dG`,
 * markers = columns [col0, col0+W), W >= H, no limit on W other than memory.  Two integer-plane products
 * (rvtests_amd/csrc/rot_gemm.hip.h) replace the symmetric block kernel.  cov[(h-col0) + (j-col0)*H] for j >= h; xz: W x d; polymorphic: W. */
int rvt_cov_rect(rvt_ctx* ctx, const double* dG, int col0, int H, int W, double* cov, double* xz, double* zz,
                 int* polymorphic);
/* The sliding window itself: a device block used as a RING — what RingMemoryPool (base/RingMemoryPool.cpp:31-63: allocate /
 * deallocate by index, nothing ever moves) is to MetaCovTest's queue of genotype vectors (src/Model.cpp:936-942, the eviction
 * loop :879-905, printCovariance :942-1004).  ring_cols > 0: logical column j of the call is PHYSICAL column
 * (col0 + j) mod ring_cols of `dG` (ring_cols <= the block's columns; the caller uploads a new site into the physical column
 * behind the tail with rvt_block_upload_columns and drops heads by advancing col0 — no column is ever moved);
 * ring_cols = 0: a linear range, columns [col0, col0 + W) as in rvt_cov_rect.
 * Heads = logical columns [0, H), markers = logical columns [0, W), H <= W (W is clipped to H + halo); halo = the most
 * markers behind a head that the caller will print (the window, in markers).  Only the BAND is computed and returned:
 *   band[h * (halo + 1) + t], t = 0 .. halo : (float)value(h, h + t) * scale — value as rvt_cov_rect's cov, the float cast
 *                          and the float multiplication by scale = (float)(1 / N) being what printCovariance applies before
 *                          "%g" (src/Model.cpp:975-984; scale = 1 gives the cast value) — NaN where h + t >= W
 *   xz[j*d + k], polymorphic[j] : the W markers;  zz as above.
 * H is not limited by RVT_MAX_VARIANTS.  Hard-call columns under an unweighted model: only the int8 copies that
 * rvt_block_upload_columns keeps are read (1 byte per genotype), the tiles of the band go to the int8 matrix cores
 * (rvtests_amd/csrc/band_gemm.hip.h); dosages / a binary trait: the same band tiles on the fp64 matrix cores.  `band` is
 * written by DMA when it lies inside a range registered with rvt_host_register.  Synchronous. */
int rvt_cov_band(rvt_ctx* ctx, const double* dG, int ring_cols, int col0, int H, int W, int halo, float scale, float* band,
                 double* xz, double* zz, int* polymorphic);
/* Which product the last rvt_cov_band call of this context took (tests, tools): 0 the fp64 matrix cores (dosages, a binary
 * trait, columns the engine knows nothing about that turned out not to be hard calls); 1 the MXFP4 band on the column cache
 * (hard calls only); 4 the MXFP4 band of MEAN-IMPUTED hard calls — every column 0 / 1 / 2 plus at most one other value known
 * from its packed upload: four exact integer products combined with the other values in fp64; 2 the MXFP4 band on a copy
 * made inside the call (no cache); 11 / 12 the same on the int8 instruction (RVT_BAND_INT8=1); -1 no call yet. */
int rvt_cov_band_last_path(rvt_ctx* ctx);
/* ---- MetaScore: single-variant score statistics (unrelated samples) ----------------------------------------------
 * Replaces the per-variant body of MetaScoreTest::fit for MetaUnrelatedQtl / MetaUnrelatedBinary
 * (src/Model.h:3246-3258 -> 3516-3549 / 3706-3769), i.e. LinearRegressionScoreTest::TestCovariate
 * (regression/LinearRegressionScoreTest.cpp:173-263) or LogisticRegressionScoreTest::TestCovariate
 * (regression/LogisticRegressionScoreTest.cpp:220-302) of ONE genotype column against the installed null model, for all
 * V columns of a device block in one pass (the block is cut into 16-column slices, each streamed once by the
 * sufficient-statistics kernel).  Outputs, V entries each, in the units MetaScoreTest::writeOutput prints:
 *   ustat = U_STAT, vstat = V (SQRT_V_STAT is its square root), effect = ALT_EFFSIZE, effect_se = ALT_EFFSIZE_SE,
 *   pvalue = PVALUE;  ok[h] = 0 for a monomorphic site (isMonomorphicMarker) or a non-positive variance, in which
 *   case the reference prints the site columns and leaves these five empty (NA).
 * Genotypes are taken as stored (imputed, NOT flipped), as MetaScoreTest does. */
int rvt_score_block(rvt_ctx* ctx, const double* dG, int V, int* ok, double* ustat, double* vstat, double* effect,
                    double* effect_se, double* pvalue);
/* What MetaScoreTest::PrintNullModel prints (src/Model.h:3526-3542, 3737-3752): the estimates beta (d; NaN when the
 * model was installed by rvt_set_null, whose caller fitted it), the diagonal of their covariance (LinearRegression
 * covB = (X'X)^-1 sigma2, regression/LinearRegression.cpp:62-66; LogisticRegression covB = (X'WX)^-1,
 * regression/LogisticRegression.cpp:330-334) and sigma2 (quantitative; 1 for a binary trait).  beta / sigma2 may be
 * NULL. */
/* dimensions of the installed null model: samples, columns of X (with the intercept) */
int rvt_null_dims(rvt_ctx* ctx, int64_t* N, int* d);
int rvt_null_summary(rvt_ctx* ctx, double* beta, double* covb_diag, double* sigma2);
/* MetaScore with kinship (MetaFamQtl, src/Model.h:3398-3499; MetaFamBinary, :3556-3668): FastLMM::TestCovariate in its
 * SCORE branch (regression/FastLMM.cpp:215-247) and FastLMM::FastGetAF (:400-424) of every raw column of a device
 * block, after rvt_set_kinship + rvt_fit_fam_null.
 *   binary = 0: genotype centred; ustat / vstat = GetUStat / GetVStat
 *   binary = 1: genotype NOT centred (disableCenterGenotype, :3558-3560); ustat = GetUStat b, vstat = GetVStat b^2 with
 *               the b of the last rvt_fam_binary_scale (:3647-3648)
 * af = the GLS allele frequency, pvalue = chisq_Q(U^2 / V, 1), 1 when V <= 0.  ALT_EFFSIZE (U / V, :3491-3494; binary
 * ustat / vstat, the same as U / V / b of :3649-3654) and its SE (1 / sqrt(V); binary 1 / sqrt(vstat) / b) are the
 * caller's divisions.  ok[h] = 0 for a monomorphic site. */
int rvt_score_block_fam(rvt_ctx* ctx, const double* dG, int V, int binary, int* ok, double* ustat, double* vstat,
                        double* af, double* pvalue);
/* Diagonal of FastLMM::GetNullCovB (regression/FastLMM.cpp:473-483) for MetaFamQtl::PrintNullModel; beta, SigmaG2 =
 * sigma2_g and SigmaE2 = sigma2_g * delta come from rvt_fit_fam_null's rvt_fam_null. */
int rvt_fam_null_summary(rvt_ctx* ctx, double* covb_diag);
/* Copy columns between two device blocks (growing the adapter's ring).  What the engine keeps per uploaded column (below)
 * travels with them when both blocks came from rvt_block_alloc. */
int rvt_block_copy_columns(rvt_ctx* ctx, double* dst, int dst_col, const double* src, int src_col, int ncols);
/* Fill columns [col0, col0+ncols) of a device block from host memory (N doubles per column, contiguous).  Behind the copy the
 * engine reads the column once on the device: it records whether the column holds hard calls only (rvt_score_block / the
 * MetaCov calls start on the integer kernels then) and, under an unweighted null model, keeps with the block what MetaCov's
 * column pass would compute for the column — its int8 copy, sum, polymorphic flag and row of T = G'X — so that rvt_cov_block /
 * rvt_cov_rect on such a block start at the integer product (same numbers, bit for bit, as for a block uploaded at once;
 * not used after the null model changed).  N more bytes of device memory per column. */
int rvt_block_upload_columns(rvt_ctx* ctx, double* dG, int col0, int ncols, const double* G);
/* Move columns [src_col, src_col+ncols) of a device block down to dst_col <= src_col (ring compaction; the per-column records
 * move with them). */
int rvt_block_move_columns(rvt_ctx* ctx, double* dG, int dst_col, int src_col, int ncols);

/* ---- null models of unrelated samples on the device ---------------------------------------------------------------
 * rvt_fit_null fits what SkatTest::fit / SkatOTest::fit / MetaCovUnrelated* fit once per analysis
 * (src/Model.h:2672-2699) and installs it as rvt_set_null would:
 *   quantitative: LinearRegression::FitLinearModel (regression/LinearRegression.cpp:20-69): beta = (X'X)^-1 X'y,
 *                 res = y - X beta, sigma2 = ||res||^2 / N, v_i = sigma2
 *   binary:       LogisticRegression::FitLogisticModel(X, y, 100) (regression/LogisticRegression.cpp:279-336): Newton /
 *                 IRLS from beta = 0, stop when rounds > 1 and |deviance change| < 1e-3, failure on a non-normal
 *                 deviance or 100 rounds; res = y - p, v = p (1 - p) with p, v of the LAST EXECUTED round (i.e. before
 *                 the final beta update, as the reference leaves them)
 * X: N x d column-major incl. intercept; y: N (0/1 for binary).  beta_out (d) and sigma2_out may be NULL.
 * Returns RVT_E_INVALID when the fit fails (singular X'VX, no convergence) — the callers then print NA rows. */
int rvt_fit_null(rvt_ctx* ctx, int trait, int64_t N, int d, const double* X, const double* y, double* beta_out,
                 double* sigma2_out);

/* ---- SKAT permutations -----------------------------------------------------------------------------------------
 * With rvt_params.skat_nperm > 0 (the reference's default for `--kernel skat`: nPerm = 10000, src/ModelManager.cpp:171-175),
 * rvt_run_blocks / rvt_collect also run the adaptive permutation test of SkatTest::fit (src/Model.h:2706-2718;
 * Permutation src/Permutation.h:69-98; permute src/LinearAlgebra.h:8-21; Skat::GetQFromNewResidual
 * regression/Skat.cpp:107-116), gene after gene.  Two modes (rvt_set_perm_exact, or RVT_PERM_EXACT=0/1 in the environment):
 *   exact (DEFAULT of a       ONE emulated glibc rand() stream consumed exactly as the reference's process-wide rand() is, so
 *   context and of a          the permutations themselves — and ActualPerm / NumGreater / NumEqual / PermPvalue — are the
 *   one-member group)         reference's (Q is evaluated in fp64 instead of fp32).  Sequential across genes.
 *   counter-based (default    shuffle s of gene g is a keyed bijection of [0, N) — Philox keys from (seed, gene_id, s), a
 *   of a group with more      cycle-walked Feistel network (rvtests_amd/csrc/perm_counter.h).  No state is shared between
 *   than one member;          genes, so any context of a device group may take any gene, in any order, and the records do
 *   rvt_set_perm_exact(c,0))  not depend on how the genes were dealt; nothing is stored per shuffle.  STATISTICAL parity
 *                             with the reference: the same estimator of the same tail probability with the same stopping
 *                             rule, other random numbers (granted to sharded runs).  ~40x faster.
 * rvt_rand_seed restarts the rand() stream (srand semantics; the reference never calls srand, i.e. seed 1, which is also
 * the state of a fresh context) and is the seed of the counter-based keys.  rvt_run_blocks_async rejects skat_nperm > 0. */
int rvt_rand_seed(rvt_ctx* ctx, unsigned seed);
int rvt_set_perm_exact(rvt_ctx* ctx, int on);

/* ---- related samples: FastLMM null + FamSKAT (`--kernel famSkat`) --------------------------------------------
 * rvt_set_kinship   installs the eigendecomposition of the kinship the caller already holds
 *                   (dc->getKinshipUForAuto() / getKinshipSForAuto(): EigenMatrix = Eigen::MatrixXf, column-major
 *                   N x N and N x 1, regression/EigenMatrix.h:9-12); kept on the device in fp64.
 * rvt_fit_fam_null  FastLMM::FitNullModel (regression/FastLMM.cpp:28-142; model MLE, test SCORE as
 *                   FamSkat.cpp:27 constructs it): rotation by U', the 101-point delta grid, the GSL-Brent
 *                   refinement of GSLMinimizer.cpp:18-66 (same evaluation sequence, so the same quirks: delta is the
 *                   bracket's minimum while beta / sigma2 belong to the last evaluated point), followed by the
 *                   products FamSkat::FitNullModel (regression/FamSkat.cpp:34-64) prepares — without forming the
 *                   N x N Sigma / Sigma^-1 / P0.  X: N x d column-major incl. intercept (d <= RVT_MAX_COV - 2), y: N.
 * rvt_run_fam_blocks  FamSkat::TestCovariate (regression/FamSkat.cpp:65-138) for n device-resident blocks (imputed,
 *                   UNFLIPPED, like rvt_run_blocks): flip-to-minor + monomorphic removal, weights
 *                   beta_pdf(FastGetAF; 1, 25) (FastLMM.cpp:402-443), Q, eigenvalues of wg P0 wg', Davies only.
 *                   Fills n_variants, n_poly, famskat_ok / famskat_Q / famskat_p of each record.  Synchronous. */
int rvt_set_kinship(rvt_ctx* ctx, int64_t N, const float* U, const float* S);
/* Structure of the installed eigenvectors.  The kinship of unrelated families is block diagonal, and then so is U: every
 * eigenvector is non-zero on one family's samples only.  rvt_set_kinship detects that (first / last non-zero row of each
 * column), re-orders the eigenpairs by support — the statistics do not depend on their order — and the rotation U'G then
 * visits only the K chunks that hold non-zeros (families listed contiguously) or gathers the few non-zeros of every
 * eigenvector (families interleaved in the sample order): visited_fraction = the share of the N x N product that is computed
 * (1 = dense U, e.g. a GRM's eigenvectors; ~4 / 782 for nuclear families at N = 100 000).  The skipped parts are exact
 * zeros, so the rotated values are those of the dense product (the statistics sum the eigenpairs in the new order: equal to
 * rounding).  RVT_KINSHIP_DENSE=1 in the environment disables the detection. */
int rvt_kinship_structure(rvt_ctx* ctx, double* visited_fraction);
int rvt_fit_fam_null(rvt_ctx* ctx, int64_t N, int d, const double* X, const double* y, rvt_fam_null* out);
int rvt_run_fam_blocks(rvt_ctx* ctx, int n_genes, const double* const* dG, const int* M, const int64_t* gene_ids,
                       rvt_gene_result* out);
/* The same with a test mask out of RVT_TEST_FAMSKAT | RVT_TEST_FAMCMC | RVT_TEST_FAMZEGGINI.  The family burden tests
 * (src/Model.h:2261-2492) collapse the flipped, polymorphic block (cmcCollapse / zegginiCollapse), and run
 * FastLMM::TestCovariate's SCORE branch (regression/FastLMM.cpp:215-247: U, V = g~' scaledK g~ / sigma2 on the centred,
 * rotated collapsed genotype, p = chisq(1)) and FastLMM::GetAF (:356-398); the collapsed columns ride through the same
 * rotation GEMM as the genotypes. */
int rvt_run_fam_tests(rvt_ctx* ctx, int n_genes, const double* const* dG, const int* M, const int64_t* gene_ids,
                      uint32_t tests, rvt_gene_result* out);
/* MetaCov with kinship, quantitative trait (MetaCovFamQtl, src/Model.cpp:437-504 over FastLMM::TransformCentered /
 * GetCovXX / GetCovXZ / GetCovZZ, regression/FastLMM.cpp:510-625): same contract as rvt_cov_block, with the null model
 * of rvt_fit_fam_null; xz is V x d, zz d x d (d = columns of X).  For the binary family variant (MetaCovFamBinary) call
 * rvt_fam_binary_scale first. */
int rvt_cov_block_fam(rvt_ctx* ctx, const double* dG, int V, double* cov, double* xz, double* zz, int* polymorphic);
/* The family counterpart of rvt_cov_rect (windows wider than one block): heads [col0, col0 + H) against markers
 * [col0, col0 + W) of the RAW block; the columns are rotated by U' on the device.  Same outputs as rvt_cov_rect (xz: W x d,
 * d = columns of X). */
int rvt_cov_rect_fam(rvt_ctx* ctx, const double* dG, int col0, int H, int W, double* cov, double* xz, double* zz,
                     int* polymorphic);
/* The family counterpart of rvt_cov_band (same arguments and outputs; xz: W x d of U'X): the ring holds RAW columns, every
 * pass of up to 1 024 heads rotates its heads and the window behind them by U'. */
int rvt_cov_band_fam(rvt_ctx* ctx, const double* dG, int ring_cols, int col0, int H, int W, int halo, float scale, float* band,
                     double* xz, double* zz, int* polymorphic);
/* MetaCovFamBinary (src/Model.cpp:595-692): after rvt_fit_fam_null on the 0/1 phenotype, scale everything
 * rvt_cov_block_fam returns by b^2, b = obtainB(alpha) = integral of logistic'(alpha + x) phi(x) dx
 * (src/Model.cpp:339-369; the reference uses gsl_integration_qagi with epsrel 1e-7), alpha = log(n_case / n_ctrl) kept
 * as float, 500 without controls.  The factor stays until the next rvt_fit_fam_null.  Outputs may be NULL. */
int rvt_fam_binary_scale(rvt_ctx* ctx, int64_t n_case, int64_t n_ctrl, double* alpha_out, double* b_out);

/* ---- KinshipHolder::decompose on the device (SURVEY §8f "next" #3) -------------------------------------------------------
 * Replaces base/KinshipHolder.cpp:270-290 (Eigen::SelfAdjointEigenSolver<MatrixXf> of the N x N kinship): K is the float
 * matrix as KinshipHolder::load fills it (column-major, symmetric), S_out (N) receives the eigenvalues in ASCENDING order
 * and U_out (N x N, column-major) the eigenvectors, as matS / matU hold them (either may be NULL).  install != 0 also
 * installs the decomposition for the family tests (as rvt_set_kinship would) without a round trip through host
 * memory.  A dense matrix of N >= 128 whose eigenvalues are simple (no two closer than 4e-9 of the spectrum's
 * width: a genetic relationship matrix) takes Householder tridiagonalisation + Sturm bisection + inverse iteration + the
 * matrix-core back-transformation (rvtests_amd/csrc/tridiag_kernels.hip.h; 20 N^2 bytes of device memory at the peak — it must be free, else the
 * matrix goes to Jacobi; sweeps = 0 in the info), closed by a check of max |K u - lambda u| and |U'U - I| on the device; anything else — repeated eigenvalues, a failed
 * check, RVT_KINSHIP_JACOBI=1 — the fp64 one-sided block Jacobi iteration (rvtests_amd/csrc/jacobi_kernels.hip.h; 16 N^2 bytes).
 * Eigenvectors of repeated eigenvalues are an arbitrary orthonormal basis of their eigenspace — exactly as for any
 * eigensolver; every statistic of the family tests depends on U only through U f(S) U'. *
 * A kinship of separate families (the connected components of its sparsity pattern, in any sample order), none larger
 * than 64 samples — the usual pedigree kinship — is recognised and decomposed family by family (the blocks are packed into
 * 64 x 64 tiles for the same Jacobi kernel; sweeps = 0 in the info): 1.7 s including the installation at N = 100 000,
 * where the dense iteration would need 200 GB and minutes.  RVT_KINSHIP_DENSE=1 forces the dense iteration. */
typedef struct rvt_decompose_info {
  int sweeps;            /* block-Jacobi sweeps (0: decomposed family by family, or through the tridiagonal form) */
  double max_cosine;     /* largest cosine between two columns of K U in the last sweep (convergence: < 1e-10) */
  int64_t padded_order;  /* order of the padded problem (multiple of 64) */
  double shift;          /* 0, or the diagonal shift of the second attempt (taken when a residual of the first was large:
                            nearly opposite eigenvalues, which a positive semi-definite kinship does not have) */
  double max_residual;   /* largest ||K u - lambda u|| */
} rvt_decompose_info;
int rvt_kinship_decompose(rvt_ctx* ctx, int64_t N, const float* K, float* U_out, float* S_out, int install,
                          rvt_decompose_info* info);

/* FamAnalyticVT (--vt famanalytic: AnalyticVT(RELATED), src/Model.h:2189-2214) after rvt_set_kinship + rvt_fit_fam_null:
 * frequencies from FastLMM::FastGetAF, (u, v) from FastLMM::CalculateUandV on the flipped, polymorphic columns of every
 * block, then the same threshold search and integral as RVT_TEST_ANALYTICVT (vt_* fields of the records).  Synchronous. */
int rvt_fam_analytic_vt(rvt_ctx* ctx, int n_genes, const double* const* dG, const int* M, rvt_gene_result* out);

/* ---- KBAC (--kernel kbac[nPerm:alpha], KBACTest src/Model.h:2891-3045 over regression/kbac.cpp) --------------------------------
 * Genotype-pattern test for BINARY traits WITHOUT covariates: every sample's multi-site genotype pattern over the
 * flipped, polymorphic columns with 0 < frequency <= 1 (imputed non-integers count as wild type), the statistic
 * sum over patterns of (case frequency - control frequency) x hypergeometric kernel weight, and nperm cumulative
 * std::random_shuffle permutations of the phenotype on the process-wide rand() stream (the same emulated stream as the
 * SKAT permutations, rvt_rand_seed; consumed in gene order) with the reference's adaptive stopping rule (every 5000
 * permutations when alpha < 1).  pvalue is what KBACTest prints with "%f".  The device holds the blocks (as for
 * rvt_run_blocks) and a null model must be set (it defines N); y = the 0 / 1 phenotype, af = the concatenated
 * frequencies dc->getMarkerFrequency(j) of every gene's columns. */
typedef struct rvt_kbac_result {
  int fit_ok;      /* 0: no polymorphic column (NA row) */
  int n_poly;      /* columns after flip-to-minor + monomorphic removal */
  int n_pattern;   /* distinct non-wild-type genotype patterns */
  int n_carrier;   /* samples with a non-wild-type pattern */
  int actual_perm; /* permuted statistics evaluated */
  int num_ge, num_le; /* permutations with statistic >= / <= observed */
  double stat;     /* observed statistic */
  double pvalue;
} rvt_kbac_result;
int rvt_kbac_blocks(rvt_ctx* ctx, int n, const double* const* dG, const int* M, const double* af, const double* y,
                    int nperm, double alpha, rvt_kbac_result* out);

/* ---- raw / packed genotypes at the boundary (SURVEY §8f "next" #1) --------------------------------------------------
 * Like rvt_submit_gene, but the block is what the genotype extractor produced, BEFORE DataConsolidator::consolidate:
 * missing genotypes are negative (-9, libVcf/VCFConstant.h:4).  The device then does what consolidate() does to the
 * genotype matrix: GenotypeCounter allele frequencies (src/GenotypeCounter.h:14-51; missing counted in the
 * denominator) and imputeGenotypeToMean with its integer-truncated allele count (src/DataConsolidator.cpp:217-245).
 *   rvt_submit_gene_raw : N x M doubles, column-major (hard calls or dosages)
 *   rvt_submit_gene_i8  : N x M int8, column-major (hard calls 0/1/2, negative = missing): 1 byte per genotype in the
 *                         caller's memory; when the gene may stay packed (see rvt_submit_gene_bed) the staging threads turn
 *                         the bytes into .bed rows on the way — a quarter of a byte per genotype over PCIe — and the gene
 *                         continues as a rvt_submit_gene_bed gene (a value above 2 is not a hard call: that gene crosses as
 *                         bytes and is expanded to doubles on the device)
 *   rvt_submit_gene_bed : PLINK .bed storage as PlinkInputFile reads it in SNP-major mode
 *                         (libVcf/PlinkInputFile.cpp:24-47, codes libVcf/PlinkInputFile.h:206-209): M rows of
 *                         ceil(N/4) bytes, sample p in bits 2(p&3).. of byte p>>2; 00 -> 0, 10 -> 1, 11 -> 2,
 *                         01 -> missing: a quarter of a byte per genotype.  Under a quantitative trait, for tests that
 *                         do not need the fp64 block (no permutations, no AnalyticVT, M <= 96), the rows stay packed on
 *                         the device and the sufficient statistics are formed from them (suffstat_hcp.hip.h): same
 *                         records, 1/64 of the device memory and traffic
 * af_out (M, may be NULL) receives the allele frequencies the tests use (dc->getMarkerFrequency). */
int rvt_submit_gene_raw(rvt_ctx* ctx, int64_t gene_id, int M, const double* Graw, uint32_t tests,
                        const rvt_params* params, double* af_out);
int rvt_submit_gene_i8(rvt_ctx* ctx, int64_t gene_id, int M, const int8_t* G8, uint32_t tests,
                       const rvt_params* params, double* af_out);
/* Several genes per call: the hand-off of rvt_submit_gene_raw (kind 1: doubles, negative = missing), rvt_submit_gene_i8
 * (kind 2) or rvt_submit_gene_bed (kind 3: PLINK 2-bit rows; kind 7: such rows resident on the device, rvt_submit_gene_bed_dev), gene after gene, without allele frequencies returned.  For
 * callers whose per-call cost (ctypes, JNI, cgo) is not negligible against the ~120 us a packed gene of N = 500 000 needs on
 * the link; the C++ adapters call the single-gene forms (src/Main.cpp:1221-1254 hands over one gene at a time). */
int rvt_submit_genes(rvt_ctx* ctx, int kind, int n, const int64_t* gene_ids, const int* M, const void* const* data,
                     uint32_t tests, const rvt_params* params);
int rvt_submit_gene_bed(rvt_ctx* ctx, int64_t gene_id, int M, const unsigned char* bed, uint32_t tests,
                        const rvt_params* params, double* af_out);
/* A .bed matrix RESIDENT in device memory (replaces, for a whole analysis, PlinkInputFile::readIntoMatrix per gene,
 * libVcf/PlinkInputFile.cpp:24-47, and the PCIe crossing of rvt_submit_gene_bed): 500 000 samples x 2 000 000 variants of 2-bit
 * codes are 250 GB — an exome-scale cohort fits the 288 GB of one MI355X, and a gene is then named by the device address of its
 * first row.
 *   rvt_bed_alloc            n_variants rows of ceil(N/4) bytes, the FILE's layout (no padding between rows; the three magic
 *                            bytes of the file are not part of it); N is the null model's sample count
 *   rvt_bed_upload           rows [first_variant, first_variant + n_variants) from host memory (returns when they are there)
 *   rvt_submit_gene_bed_dev  like rvt_submit_gene_bed with d_rows = d_bed + first_row * ceil(N/4): M consecutive rows.  The
 *                            rows are read when the gene is computed: they must stay unchanged until its record has been
 *                            collected.  Same records as rvt_submit_gene_bed of the same rows to 1e-11 relative (SKAT-O's p to
 *                            1e-6): resident genes form G'[X | rr] on the int8 matrix cores from digit planes of the null
 *                            tile; allele frequencies, and genes that have to be expanded, bit for bit.
 *   rvt_submit_genes         kind 7 = the same, several genes per call (data[g] = device address of gene g's first row)
 *   rvt_bed_free             waits for queued genes, then frees */
int rvt_bed_alloc(rvt_ctx* ctx, int64_t n_variants, unsigned char** d_bed);
int rvt_bed_upload(rvt_ctx* ctx, unsigned char* d_bed, int64_t first_variant, int64_t n_variants, const unsigned char* rows);
int rvt_bed_free(rvt_ctx* ctx, unsigned char* d_bed);
int rvt_submit_gene_bed_dev(rvt_ctx* ctx, int64_t gene_id, int M, const unsigned char* d_rows, uint32_t tests,
                            const rvt_params* params, double* af_out);
/* Single-variant score tests of V consecutive rows of a resident .bed matrix: rvt_score_block's outputs (the per-variant body of
 * MetaScoreTest::fit, i.e. LinearRegressionScoreTest::TestCovariate on one column, regression/LinearRegressionScoreTest.cpp:
 * 173-263) computed from the 2-bit rows where they lie — N/4 bytes of device traffic per site instead of 8 N — with missing calls
 * imputed to the column mean as DataConsolidator does (src/DataConsolidator.cpp:217-245).  counts (optional, 4 V entries):
 * per variant the numbers of 0 / 1 / 2 / missing calls, what the adapter's AF / call-rate / HWE columns are made of
 * (src/Model.h:3211-3230).  Quantitative traits (the packed-row kernel's domain; RVT_E_STATE otherwise).  Synchronous. */
int rvt_score_bed_dev(rvt_ctx* ctx, const unsigned char* d_rows, int64_t V, int* ok, double* ustat, double* vstat, double* effect,
                      double* effect_se, double* pvalue, long long* counts);

/* ---- VCF text at the boundary (SURVEY §8f "next" #1: the genotype front end) -------------------------------------------
 * Replaces the per-sample loop of VCFGenotypeExtractor::extractMultipleGenotype (src/VCFGenotypeExtractor.cpp:29-140) for
 * hard calls: the caller hands over the TEXT of the sample columns of each record of a gene; splitting at tabs and ':',
 * VCFValue::getGenotype (libVcf/VCFValue.h:74-117; '.', multi-allelic or malformed calls -> missing, haploid calls -> 0 / 1),
 * the GD / GQ filters (src/VCFGenotypeExtractor.cpp:304-317) and then everything rvt_submit_gene_i8 does (allele
 * frequencies, mean imputation) run on the device.
 *   rvt_vcf_locate      host-only helper: offset of the first sample column of a record line and the FORMAT indices
 *                       of GT / GD / GQ by VCFRecord::getFormatIndex's prefix rule (libVcf/VCFRecord.h:280-305); -1 = absent
 *   rvt_vcf_set_samples once per file: row_of_sample[s] = row of the analysis (0 .. N-1) that sample column s of the
 *                       file feeds, -1 = sample not analysed (VCFPeople include / exclude); every row exactly once
 *   rvt_vcf_set_filters GDmin / GDmax / GQmin / GQmax, <= 0 = off
 *   rvt_submit_gene_vcf per record j: sample_text[j] = first byte of the first sample column, text_len[j] = bytes up to
 *                       (not including) the end of line, gt_index[j] (and gd_index / gq_index, may be NULL) from
 *                       rvt_vcf_locate.  The text is consumed before the call returns.  A record whose column count
 *                       differs from the file's sample count voids THIS gene: with af_out the call returns
 *                       RVT_E_INVALID, without it the gene's record comes back with RVT_ST_INPUT_ERROR (every test NA)
 *                       and rvt_last_error names the record.  Other genes are unaffected. */
int rvt_vcf_locate(const char* line, int64_t len, int64_t* sample_off, int* gt_index, int* gd_index, int* gq_index);
int rvt_vcf_set_samples(rvt_ctx* ctx, int n_file_samples, const int32_t* row_of_sample);
int rvt_vcf_set_filters(rvt_ctx* ctx, int gd_min, int gd_max, int gq_min, int gq_max);
/* --dosage TAG (useDosage, src/VCFGenotypeExtractor.cpp:66-74,403-414): with use_dosage != 0 the index handed to
 * rvt_submit_gene_vcf as gt_index is the FORMAT index of the dosage tag (rvt_vcf_format_index) and every value is
 * atof() of that subfield (VCFValue::toDouble; a missing subfield or "." reads 0.0, as in the reference) — doubles, so
 * the gene then follows the path of rvt_submit_gene_raw.  Numbers of up to 15 significant digits with a decimal exponent
 * within +-22 are rounded exactly as strtod rounds them; anything else is reported as an error, never approximated. */
int rvt_vcf_set_dosage(rvt_ctx* ctx, int use_dosage);
/* --multipleAllele (multiAllelicMode, src/VCFGenotypeExtractor.cpp:44-47,441-484): the caller submits a record once per
 * alternative allele; alt[j] > 0 makes record j of the NEXT rvt_submit_gene_vcf / rvt_vcf_decode call count that allele
 * (VCFValue::countAltAllele, libVcf/VCFValue.h:180-213), alt[j] = 0 keeps the bi-allelic coding.  Consumed by that call. */
int rvt_vcf_set_alt_alleles(rvt_ctx* ctx, int M, const int* alt);
/* Hemizygous regions (--xHemi / ParRegion::isHemiRegion, src/VCFGenotypeExtractor.cpp:85-86,416-426): sex[s] = PLINK code
 * of FILE sample s (1 male, 2 female, anything else unknown; once per file, after rvt_vcf_set_samples); hemi[j] != 0 marks
 * record j of the NEXT rvt_submit_gene_vcf / rvt_vcf_decode call as lying in a hemizygous region (the caller evaluates
 * ParRegion on chrom:pos).  There a male is coded by VCFValue::getMaleNonParGenotype02 (haploid "1" or homozygous "1/1"
 * = 2, heterozygous = missing; libVcf/VCFValue.h:125-142) or countMaleNonParAltAllele2 in multi-allelic mode, a female
 * as everywhere else, unknown sex is missing; in dosage mode a male's value is doubled.  Consumed by that call. */
int rvt_vcf_set_sex(rvt_ctx* ctx, int n_file_samples, const int8_t* sex);
int rvt_vcf_set_hemi(rvt_ctx* ctx, int M, const int* hemi);
int rvt_vcf_format_index(const char* line, int64_t len, const char* key, int* index);
int rvt_vcf_decode_dosage(rvt_ctx* ctx, int M, const char* const* sample_text, const int64_t* text_len,
                          const int* tag_index, const int* gd_index, const int* gq_index, double* out);
int rvt_submit_gene_vcf(rvt_ctx* ctx, int64_t gene_id, int M, const char* const* sample_text, const int64_t* text_len,
                        const int* gt_index, const int* gd_index, const int* gq_index, uint32_t tests,
                        const rvt_params* params, double* af_out);
/* decode only (synchronous): out = N x M signed bytes, column-major, N = rows of the sample map; 0 / 1 / 2, missing = -9
 * — VCFGenotypeExtractor's matrix before consolidate(), one byte per genotype */
int rvt_vcf_decode(rvt_ctx* ctx, int M, const char* const* sample_text, const int64_t* text_len, const int* gt_index,
                   const int* gd_index, const int* gq_index, int8_t* out);

/* ---- BGEN genotype-probability blocks at the boundary (SURVEY §8f "next" #1: "also enables BGEN ... input") ------------------
 * Replaces, for the UNCOMPRESSED probability block of each variant of a gene (what the reference holds after uncompress /
 * ZSTD_decompress, libBgen/BGenFile.cpp:289-319 — the caller inflates; variant identifying data and the .bgi index stay
 * with the caller):
 *   BGenFile::parseLayout1 (libBgen/BGenFile.cpp:205-238; v1.1: 3 x uint16 / 32768, an all-zero triple = missing),
 *   BGenFile::parseLayout2 + BitReader (libBgen/BGenFile.cpp:321-392, libBgen/BitReader.h: v1.2 / v1.3 — per-sample
 *     ploidy / missing bytes, phased or unphased, any number of alleles, 1..32-bit values, float(v) * scale and the float
 *     remainder 1 - sum),
 *   BGenGenotypeExtractor::getGenotype (src/BGenGenotypeExtractor.cpp:413-478: always a dosage, prob[1] + 2 prob[2] of the
 *     sample's stored probabilities — kept as written also for phased and haploid samples —, 2 for a single allele, the
 *     normalised form for more than two alleles, MISSING_GENOTYPE for a missing sample or a ploidy other than 1 / 2)
 * and then everything rvt_submit_gene_raw does (GenotypeCounter frequencies, mean imputation) — all on the device, with the
 * reference's arithmetic type at every step: the doubles are the reference's bit for bit.  layout = 1 or 2 (the file
 * header's flag, one per file).  The sample map of rvt_vcf_set_samples applies (file sample -> analysis row, -1 = not
 * analysed); without one, file sample i is row i.  multiAllelicMode: rvt_vcf_set_alt_alleles applies to the next call —
 * the reference supports the first alternative allele only (getGenotypeForAltAllele, :470-482: alt > 1 -> every sample
 * missing).  The hemizygous-region / sex checks of getGenotype only log and have no counterpart.  A block shorter than its ploidy bytes demand is reported by a later
 * submit (RVT_E_INVALID), like a malformed VCF record. */
int rvt_submit_gene_bgen(rvt_ctx* ctx, int64_t gene_id, int M, const unsigned char* const* block, const int64_t* block_len,
                         int layout, uint32_t tests, const rvt_params* params, double* af_out);
/* decode only (synchronous): out = n_rows x M doubles, column-major; the genotype before consolidate(), -9 = missing */
int rvt_bgen_decode(rvt_ctx* ctx, int M, const unsigned char* const* block, const int64_t* block_len, int layout,
                    int64_t n_rows, double* out);

/* ---- device groups: several GPUs of one node behind one calling thread ---------------------------------------------------
 * Genes are independent units that share only the null model, so a group is one engine context per device
 * (rvtests_amd/csrc/rvt_group.cpp): the null model / kinship decomposition is installed on every member, the gene stream is
 * dealt to the members in runs of 32 genes, and rvt_group_collect returns the records in SUBMISSION order whichever member
 * produced them (the reference's output files are gene-ordered, src/Main.cpp:1221-1254).  There is no device-to-device
 * traffic; the ordered merge of the fixed-size records on the host is the only "gather".  A gene that asks for
 * permutation p-values goes to member 0 (one process-wide random stream, src/Permutation.h:69-98).  The same device may
 * be listed more than once (testing).  dev_ids == NULL: devices 0 .. n_dev-1. */
typedef struct rvt_group rvt_group;
int rvt_group_init(rvt_group** group, int n_dev, const int* dev_ids);
void rvt_group_destroy(rvt_group* group);
int rvt_group_size(const rvt_group* group);
rvt_ctx* rvt_group_member(rvt_group* group, int k);
const char* rvt_group_last_error(const rvt_group* group);
int rvt_group_set_null(rvt_group* group, int trait, int64_t N, int d, const double* X, const double* res, const double* v,
                       double sigma2);
int rvt_group_fit_null(rvt_group* group, int trait, int64_t N, int d, const double* X, const double* y, double* beta_out,
                       double* sigma2_out);
int rvt_group_submit_gene(rvt_group* group, int64_t gene_id, int M, const double* G, const double* af, uint32_t tests,
                          const rvt_params* params);
int rvt_group_submit_gene_raw(rvt_group* group, int64_t gene_id, int M, const double* Graw, uint32_t tests,
                              const rvt_params* params, double* af_out);
int rvt_group_submit_gene_i8(rvt_group* group, int64_t gene_id, int M, const int8_t* G8, uint32_t tests,
                             const rvt_params* params, double* af_out);
int rvt_group_submit_gene_bed(rvt_group* group, int64_t gene_id, int M, const unsigned char* bed, uint32_t tests,
                              const rvt_params* params, double* af_out);
int rvt_group_vcf_set_samples(rvt_group* group, int n_file_samples, const int32_t* row_of_sample);
int rvt_group_vcf_set_filters(rvt_group* group, int gd_min, int gd_max, int gq_min, int gq_max);
int rvt_group_submit_gene_vcf(rvt_group* group, int64_t gene_id, int M, const char* const* sample_text,
                              const int64_t* text_len, const int* gt_index, const int* gd_index, const int* gq_index,
                              uint32_t tests, const rvt_params* params, double* af_out);
int rvt_group_submit_gene_bgen(rvt_group* group, int64_t gene_id, int M, const unsigned char* const* block,
                               const int64_t* block_len, int layout, uint32_t tests, const rvt_params* params,
                               double* af_out);
/* SKAT permutations in a group: a ONE-member group replays the reference's rand() stream (exact, as a single context);
 * a group that deals genes over several members takes the counter-based permutations (any member takes any gene).
 * rvt_group_set_perm_exact(g, 1) sends every permutation gene to member 0, which replays the reference's rand() stream;
 * rvt_group_set_perm_exact(g, 0) selects the counter-based mode for a one-member group too (see rvt_set_perm_exact) */
int rvt_group_set_perm_exact(rvt_group* group, int on);
int rvt_group_rand_seed(rvt_group* group, unsigned seed);
/* rvt_host_register / rvt_host_unregister for a group: the range is page-locked once, for every member's device */
int rvt_group_host_register(rvt_group* group, const void* ptr, size_t bytes);
int rvt_group_host_unregister(rvt_group* group, const void* ptr);
/* rvt_set_content_hint + rvt_set_dosage_lattice on every member (what the caller's blocks hold: see those functions) */
int rvt_group_set_content(rvt_group* group, int hint, int lattice_denominator);
/* rvt_set_dosage_float on every member */
int rvt_group_set_dosage_float(rvt_group* group, int on);
/* `--meta score` / `--meta cov` over a group (chunks with a one-window halo, no exchange).  G_host: N x V
 * column-major (leading dimension N), the genotype vectors of V consecutive single-variant fit() calls; one host thread per
 * member for the duration of the call.
 *   rvt_group_score_block_host  outputs as rvt_score_block (V entries each); the columns are cut into one share per member
 *   rvt_group_cov_band_host     band[h * (halo + 1) + t] = the value of head h and marker h + t as rvt_cov_rect gives it
 *                               (t = 0 .. halo; NaN beyond the last variant); heads are dealt in chunks (chunk = 0: a
 *                               default) to the members in turn, every chunk's device block holds its heads and the
 *                               `halo` columns behind them.  xz: V x d, zz: d x d (may be NULL), polymorphic: V. */
int rvt_group_score_block_host(rvt_group* group, int64_t N, int V, const double* G_host, int* ok, double* ustat,
                               double* vstat, double* effect, double* effect_se, double* pvalue);
int rvt_group_cov_band_host(rvt_group* group, int64_t N, int V, const double* G_host, int halo, int chunk, double* band,
                            double* xz, double* zz, int* polymorphic);
int rvt_group_collect(rvt_group* group, rvt_gene_result* out, int cap, int* n_out);
int rvt_group_collect_ready(rvt_group* group, rvt_gene_result* out, int cap, int* n_out); /* cf. rvt_collect_ready */
/* related samples: the kinship decomposition is replicated on every member (6 N^2 bytes each); rvt_group_run_fam_tests_host
 * deals HOST genotype blocks (N x M[g] doubles, imputed, unflipped) to the members in contiguous shares balanced by column
 * count, every member uploads and runs its share concurrently (one host thread per member inside the call). */
int rvt_group_set_kinship(rvt_group* group, int64_t N, const float* U, const float* S);
int rvt_group_fit_fam_null(rvt_group* group, int64_t N, int d, const double* X, const double* y, rvt_fam_null* out);
int rvt_group_run_fam_tests_host(rvt_group* group, int n_genes, const double* const* G_host, const int* M,
                                 const int64_t* gene_ids, uint32_t tests, rvt_gene_result* out);

/* ---- test / inspection hooks ---------------------------------------------------------------------- */
/* collapsed burden vectors of ONE block (bit-exact parity checks): cmc_out/zeg_out are host N-vectors */
int rvt_debug_collapse(rvt_ctx* ctx, const double* dG, int M, double* cmc_out, double* zeg_out,
                       int* flipped_out, int* kept_out);
/* sufficient statistics of ONE block as the MFMA kernel produced them: S (M x M), T (M x d), u (M),
 * colsum/min/max (M each); host outputs, row-major */
int rvt_debug_suffstat(rvt_ctx* ctx, const double* dG, int M, double* S, double* T, double* u, double* colsum,
                       double* cmin, double* cmax);
int rvt_set_profiling(rvt_ctx* ctx, int on);
int rvt_get_timing(rvt_ctx* ctx, rvt_timing* t, int reset);
/* the HIP stream (hipStream_t) the engine launches on, for callers that record their own events */
void* rvt_stream(rvt_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RVTESTS_AMD_H_ */
