/* ORACLE/_ref — TEST INFRASTRUCTURE ONLY.
 * The reference's own multivariate-normal code: regression/libMvtnorm/mvt.f (Genz's MVTDST, Fortran 77) compiled where
 * it lies with the ROCm image's flang, plus its uniform source regression/libMvtnorm/randomF77.c (rand() / RAND_MAX).
 * This file only adds the call MvtNorm::compute_Band makes (regression/libMvtnorm/mvtnorm.cpp:27-47,57-72,123-151:
 * nu = 0, maxpts = 25000, abseps = 0.001, releps = 0, INFIN = 2 for every limit, DELTA = 0).
 * Output: oracle/_ref/libref_mvt.so (git-ignored). */
#include <stdlib.h>

extern void mvtdst_(int* n, int* nu, double* lower, double* upper, int* infin, double* correl, double* delta, int* maxpts,
                    double* abseps, double* releps, double* error, double* value, int* inform);

/* correl: packed strict lower triangle (2,1), (3,1), (3,2), ...  Returns inform; *prob, *err as MVTDST reports them. */
int ref_mvn_band(int n, double T, const double* correl, unsigned seed, double* prob, double* err) {
  int nu = 0, maxpts = 25000, inform = 0, i;
  double abseps = 0.001, releps = 0.0, error = 0.0, value = 0.0;
  double* lower = (double*)malloc(sizeof(double) * n);
  double* upper = (double*)malloc(sizeof(double) * n);
  double* delta = (double*)calloc(n, sizeof(double));
  int* infin = (int*)malloc(sizeof(int) * n);
  double* cor = (double*)malloc(sizeof(double) * (n > 1 ? n * (n - 1) / 2 : 1));
  for (i = 0; i < n; ++i) {
    lower[i] = -T;
    upper[i] = T;
    infin[i] = 2;
  }
  for (i = 0; i < n * (n - 1) / 2; ++i) cor[i] = correl[i];
  srand(seed);
  mvtdst_(&n, &nu, lower, upper, infin, cor, delta, &maxpts, &abseps, &releps, &error, &value, &inform);
  *prob = value;
  *err = error;
  free(lower);
  free(upper);
  free(delta);
  free(infin);
  free(cor);
  return inform;
}
