// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// C API of the CPU restatement of the rvtests kernel/burden hot path (see each .cpp for the
// reference file:line it follows).  Loaded with ctypes by tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py — never by rvtests_amd/ itself.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef double (*orc_integrand)(double x, void* params);

/* ---- scalar special functions (orc_special.cpp) ---- */
double orc_gamma_inc_P(double a, double x);
double orc_gamma_inc_Q(double a, double x);
double orc_gamma_cdf_P(double x, double a, double b);
double orc_gamma_cdf_Q(double x, double a, double b);
double orc_gamma_cdf_Qinv(double Q, double a, double b);
double orc_gamma_pdf(double x, double a, double b);
double orc_chisq_P(double x, double nu);
double orc_chisq_Q(double x, double nu);
double orc_chisq_Qinv(double Q, double nu);
double orc_chisq_pdf(double x, double nu);
double orc_beta_pdf(double x, double a, double b);

/* ---- Davies / Liu (orc_davies.cpp, orc_liu.cpp) ---- */
double orc_qf(const double* lb, const double* nc, const int* n, int r, double sigma, double c, int lim,
              double acc, double* trace, int* ifault);
double orc_davies_pvalue(const double* lambda, int n, double Q, int* fault_out);
double orc_liu_pvalue(const double* lambda, int n, double Q);
void orc_cumchn(double x, double df, double pnonc, double* cum, double* ccum);

/* ---- QAGS (orc_qags.cpp) ---- */
int orc_qags(orc_integrand f, void* params, double a, double b, double epsabs, double epsrel, int limit,
             double* result, double* abserr, int* neval_out);
/* test hook: integrate one of a few built-in integrands (id) with parameter alpha */
int orc_qags_builtin(int id, double alpha, double a, double b, double epsabs, double epsrel, int limit,
                     double* result, double* abserr, int* neval_out);

/* ---- dense helpers (orc_linalg.cpp) ---- */
/* eigenvalues (ascending) of the symmetric n x n column-major matrix A (cyclic Jacobi) */
void orc_sym_eigvals(const double* A, int n, double* w);

/* ---- null models (orc_models.cpp) ---- */
/* LinearRegression::FitLinearModel: X is N x d column-major incl. intercept. Outputs beta[d],
   predicted[N], resid[N], sigma2 (= RSS/N).  Returns 0 on success. */
int orc_fit_linear(const double* X, const double* y, int64_t N, int d, double* beta, double* pred,
                   double* resid, double* sigma2);
/* LogisticRegression::FitLogisticModel(X, y, nrrounds): outputs beta[d], p[N], v[N]. 0 = ok, -1 = failed */
int orc_fit_logistic(const double* X, const double* y, int64_t N, int d, int nrrounds, double* beta,
                     double* p, double* v);

/* ---- DataConsolidator semantics ---- */
/* imputeGenotypeToMean in place (G: N x M col-major, missing < 0). */
void orc_impute_mean(double* G, int64_t N, int M);
/* GenotypeCounter AF per column of the raw (un-imputed) matrix. */
void orc_counter_af(const double* Graw, int64_t N, int M, double* af);
/* getFlippedToMinorPolymorphicGenotype: writes N x Mout matrix, returns Mout; flipped[M], kept[M] flags */
int orc_flip_poly(const double* G, int64_t N, int M, double* out, int* flipped, int* kept);

/* ---- per-gene tests.  G is the imputed, UNFLIPPED N x M block (as dc->getGenotype()),
        af[M] the counter AF (quirk #3: indexed by filtered column), X incl. intercept. ---- */
typedef struct {
  int fit_ok;      /* 1 if the reference would print numbers, 0 for NA */
  int n_poly;      /* columns after flip + monomorphic removal */
  double Q;
  double pvalue;
  double rho;      /* SKAT-O only */
  int n_lambda;    /* SKAT: kept eigenvalues */
  double lambda[512];
  /* SKAT-O diagnostics */
  double Qs[11], pvals[11], taus[11], qminp[11];
  double muQ, varQ, varZeta, df, minP;
  int qags_status, qags_neval;
} orc_kernel_result;

typedef struct {
  int fit_ok;
  int n_poly;
  int nonref_site; /* CMC only */
  double U, V, stat, pvalue;
} orc_burden_result;

/* SkatTest::fit + Skat::Fit with the P0 projection folded (fp64).  binary: 0 QT, 1 binary. */
int orc_skat(const double* G, const double* af, int64_t N, int M, const double* X, int d, const double* res,
             const double* v, int binary, double beta1, double beta2, orc_kernel_result* out);
/* literal Skat::Fit with the N x N P0 (fp64 arithmetic; use_float=1 mirrors the reference's float casts) */
int orc_skat_literal(const double* G, const double* af, int64_t N, int M, const double* X, int d,
                     const double* res, const double* v, int binary, double beta1, double beta2, int use_float,
                     orc_kernel_result* out);
/* SkatOTest::fit + SkatO::Fit, literal operation order (Z1, Z1*L, Z2'Z2 per rho). */
int orc_skato(const double* G, const double* af, int64_t N, int M, const double* X, int d, const double* res,
              const double* v, int binary, double beta1, double beta2, orc_kernel_result* out);
/* MetaScoreTest (unrelated samples): per-column score statistics in the printed units + the null-model summary. */
int orc_metascore(const double* G, int64_t N, int V, const double* X, int d, const double* y, int binary, int* ok,
                  double* ustat, double* vstat, double* effect, double* se, double* pval, double* beta_out,
                  double* covb_diag, double* sigma2_out);
/* CMCTest / ZegginiTest: which = 0 CMC, 1 Zeggini.  y and X are used to refit the null as the reference does. */
int orc_burden(const double* G, int64_t N, int M, const double* X, int d, const double* y, int binary, int which,
               orc_burden_result* out);
/* collapsed vectors only (bit-exact checks) */
void orc_collapse(const double* G, int64_t N, int M, int which, double* out);

/* Permutation p-value machinery (src/Permutation.h:69-98, src/LinearAlgebra.h:8-21) with an explicit
   glibc-TYPE_3 rand() emulator state so that runs are reproducible. */
typedef struct {
  int num_perm, actual_perm, num_x, num_equal;
  double threshold, pvalue;
} orc_perm_result;
void orc_rand_seed(unsigned seed);
int orc_rand(void);
int orc_skat_permute(const double* G, const double* af, int64_t N, int M, const double* res, double beta1,
                     double beta2, double obs, int nPerm, double alpha, int use_float, orc_perm_result* out);

/* ---- related samples (orc_fam.cpp): FastLMM null model (MLE, score) and FamSKAT ----
        U: N x N column-major eigenvectors of the kinship, S: N eigenvalues (EigenMatrix holds them as float; here
        doubles carrying those values).  use_float = 1 restates the reference's fp32 arithmetic. */
typedef struct {
  int ok;
  int max_index;    /* best point of the 101-point delta grid */
  int brent_evals;  /* goal-function evaluations inside Minimizer::minimize (0 on the boundary) */
  double delta;     /* sigma2_e / sigma2_g as FastLMM::GetDelta() returns it */
  double sigma2;    /* FastLMM::GetSigmaG2() */
  double beta[16];  /* FastLMM::GetBeta() */
} orc_fam_null;
/* The Brent minimiser restatement (GSL 1.16 min/brent.c as driven by Minimizer::minimize, GSLMinimizer.cpp:18-66) on
   built-in test functions, for the fixture check against GSL itself:
     id 0: (x-a)^2   1: cosh(x-a)   2: x^4 - a x   3: -log-likelihood-like  a/x + log(x)   4: |x-a|^1.5
   Returns the minimize() status (0 / -1); outputs x_minimum, number of function evaluations, last evaluated x. */
int orc_brent_builtin(int id, double a, double start, double lb, double ub, double* xmin, int* evals, double* last_x);
/* FastLMM::FitNullModel (regression/FastLMM.cpp:28-142) incl. the GSL Brent refinement (GSLMinimizer.cpp:18-66) */
int orc_fastlmm_null(const double* X, const double* y, int64_t N, int d, const double* U, const double* S,
                     int use_float, orc_fam_null* out);
/* FamSkat::FitNullModel + TestCovariate (regression/FamSkat.cpp:34-138) with the literal N x N Sigma / P0;
   G is the imputed UNFLIPPED block (the function applies getFlippedToMinorPolymorphicGenotype itself).
   Fills fit_ok, n_poly, Q, pvalue, n_lambda, lambda.  Returns -1 when no polymorphic column is left. */
int orc_famskat(const double* G, int64_t N, int M, const double* X, const double* y, int d, const double* U,
                const double* S, const orc_fam_null* nul, int use_float, orc_kernel_result* out);

/* FamCMC / FamZeggini (src/Model.h:2261-2492): cmcCollapse / zegginiCollapse of the flipped, polymorphic block, then
   FastLMM::TestCovariate in its SCORE branch (regression/FastLMM.cpp:215-247, scaledK LITERALLY as an N x N matrix) and
   FastLMM::GetAF (:356-398).  which: 0 = CMC, 1 = Zeggini, 2 = MetaScoreTest's MetaFamQtl (src/Model.h:3421-3434: the
   single raw column itself, M == 1, no flip), 3 = MetaFamBinary (as 2 with the genotype NOT centred, :3558-3560; the
   caller applies the b scaling of :3647-3662).  Returns -1 when no polymorphic column is left. */
typedef struct {
  int fit_ok, num_site;
  double af, U, V, stat, pvalue;
} orc_fam_burden_result;
int orc_fam_burden(const double* G, int64_t N, int M, const double* X, const double* y, int d, const double* U,
                   const double* S, const orc_fam_null* nul, int which, int use_float, orc_fam_burden_result* out);
/* FastLMM::GetNullCovB (regression/FastLMM.cpp:473-483), d x d. */
int orc_fastlmm_covb(const double* X, int64_t N, int d, const double* U, const double* S, double delta, int use_float,
                     double* covb);
/* obtainB (src/Model.cpp:339-369): b = integral over R of logistic'(alpha + x) phi(x) dx, evaluated by the reference
   with gsl_integration_qagi (epsrel 1e-7).  Restated with the QAGS restatement on QAGI's own change of variable
   x = (1 - t) / t over (0, 1] (both half lines folded); the integrand is smooth, so the two adaptive rules agree far
   below the 6 digits that are printed. */
double orc_obtain_b(double alpha);
/* MetaCovTest with kinship, quantitative trait (MetaCovFamQtl, src/Model.cpp:437-504 over FastLMM::TransformCentered /
   GetCovXX / GetCovXZ / GetCovZZ, regression/FastLMM.cpp:510-625): same outputs as orc_metacov. */
int orc_metacov_fam(const double* G, int64_t N, int V, const int* chrom, const int* pos, const double* X, int d,
                    const double* U, const double* S, const orc_fam_null* nul, int window, int use_float, int* kept,
                    double* cov, int* row_end, double* xz, double* zz);
/* MetaCovFamBinary (src/Model.cpp:595-692): the same with covXX, covXZ, covZZ scaled by b^2, b = obtainB(alpha),
   alpha = log(nCase / nCtrl) stored as float (500 when there is no control); y is the 0/1 phenotype the FastLMM null
   (`nul`) was fitted to. */
int orc_metacov_fam_binary(const double* G, int64_t N, int V, const int* chrom, const int* pos, const double* X, int d,
                           const double* y, const double* U, const double* S, const orc_fam_null* nul, int window,
                           int use_float, int* kept, double* cov, int* row_end, double* xz, double* zz);

/* ---- MetaCovTest for unrelated samples (src/Model.cpp:844-1004; MetaCovUnrelatedQtl :506-593,
        MetaCovUnrelatedBinary :694-778; window rule src/Model.h:3956-3990).
        G: N x V imputed genotypes (one variant per column, file order); chrom[V] (any integer id), pos[V];
        X: N x d incl. intercept; y: N.  use_float = 1 restates the reference's fp32 storage/arithmetic.
        Outputs (all caller-allocated):
          kept[V]      1 if the variant entered the queue (not monomorphic)
          cov[V*V]     cov[h + j*V] for j >= h in the same window = the value printed for (head h, marker j)
                       BEFORE the 1/N scaling; NaN elsewhere
          row_end[V]   for a kept head h: index of the last marker of its output row; -1 if not kept
          xz[V*d]      cov(G,Z) of each kept variant (row-major by variant), unscaled
          zz[d*d]      covZZ, unscaled
        Returns 0, or -1 if the null model cannot be fitted. ---- */
int orc_metacov(const double* G, int64_t N, int V, const int* chrom, const int* pos, const double* X, const double* y,
                int d, int binary, int window, int use_float, int* kept, double* cov, int* row_end, double* xz,
                double* zz);

#ifdef __cplusplus
}
#endif
