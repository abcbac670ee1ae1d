#!/usr/bin/env python3
"""Generate tests/golden/gsl_brent.json from GSL 1.16 (third/gsl-1.16.tar.gz, the library the reference links):
gsl_min_fminimizer_brent driven exactly as the reference's Minimizer::minimize does (regression/GSLMinimizer.cpp:18-66:
set; iterate; stop when gsl_min_test_interval(a, b, 1e-3, 0) succeeds or after 100 iterations), on a few test
functions.  Records x_minimum, the number of function evaluations and the LAST evaluated abscissa (FastLMM's beta /
sigma2 are side effects of that evaluation, regression/FastLMM.cpp:812-817).

Run ONLY in the build container after building GSL as make_gsl_golden.py describes.  The fixture is data."""
import json
import os
import subprocess
import tempfile

GSL = os.environ.get("GSL_PREFIX", "/tmp/gslinst")
HERE = os.path.dirname(os.path.abspath(__file__))

C_SRC = r'''
#include <stdio.h>
#include <math.h>
#include <gsl/gsl_errno.h>
#include <gsl/gsl_min.h>
static int n_eval; static double last_x; static int fid; static double fa;
static double f(double x, void* p){ ++n_eval; last_x = x;
  switch(fid){ case 0: return (x-fa)*(x-fa); case 1: return cosh(x-fa); case 2: return x*x*x*x - fa*x;
               case 3: return fa/x + log(x); default: return pow(fabs(x-fa),1.5);} }
int main(void){
  gsl_set_error_handler_off();
  double start, lb, ub;
  while (scanf("%d %lf %lf %lf %lf", &fid, &fa, &start, &lb, &ub) == 5) {
    gsl_function F; F.function = f; F.params = 0; n_eval = 0; last_x = NAN;
    gsl_min_fminimizer* s = gsl_min_fminimizer_alloc(gsl_min_fminimizer_brent);
    int status = gsl_min_fminimizer_set(s, &F, start, lb, ub), rc = 0, iter = 0; double x = start;
    if (status != GSL_SUCCESS) rc = -1;
    else do { iter++; status = gsl_min_fminimizer_iterate(s);
              if (status == GSL_EBADFUNC || status == GSL_FAILURE) { rc = -1; break; }
              x = gsl_min_fminimizer_x_minimum(s);
              status = gsl_min_test_interval(gsl_min_fminimizer_x_lower(s), gsl_min_fminimizer_x_upper(s), 0.001, 0.0);
              if (status == GSL_SUCCESS) break; } while (status == GSL_CONTINUE && iter < 100);
    printf("%d %.17g %d %.17g\n", rc, x, n_eval, last_x);
    gsl_min_fminimizer_free(s);
  }
  return 0; }
'''


def main():
    cases = []
    for a in (0.3, 1.7, 2.5, 0.011):
        cases += [(0, a, a * 1.1 + 0.01, a - 1.0, a + 2.0), (1, a, a + 0.3, a - 2.0, a + 1.0),
                  (3, a, a * 0.9, a * 0.5, a * 3.0), (4, a, a + 0.2, a - 1.0, a + 1.5)]
    cases += [(2, 4.0, 0.9, 0.0, 2.0), (2, 32.0, 1.5, 0.5, 3.0),
              (0, 1.0, 0.5, 0.0, 0.8),            # minimum outside: set() rejects the bracket
              (3, 0.0007, 0.0007 * 1.05, 0.0007 * 0.8187, 0.0007 * 1.2214),   # bracket already < 1e-3
              (3, 0.72, 0.72, 0.72 * 0.8187307531, 0.72 * 1.2214027582)]      # FastLMM-like grid bracket
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "b.c"), os.path.join(td, "b")
        open(src, "w").write(C_SRC)
        subprocess.check_call(["gcc", "-O2", "-I" + GSL + "/include", src, "-o", exe, "-L" + GSL + "/lib", "-lgsl",
                               "-lgslcblas", "-lm"])
        inp = "\n".join("%d %r %r %r %r" % c for c in cases) + "\n"
        out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    rows = []
    for c, line in zip(cases, out):
        rc, x, n, last = line.split()
        rows.append({"id": c[0], "a": c[1], "start": c[2], "lb": c[3], "ub": c[4], "rc": int(rc), "xmin": float(x),
                     "evals": int(n), "last_x": float(last)})
    json.dump({"source": "GSL 1.16 gsl_min_fminimizer_brent, epsabs 1e-3, epsrel 0, <= 100 iterations", "cases": rows},
              open(os.path.join(HERE, "gsl_brent.json"), "w"), indent=1)
    print("wrote", len(rows), "cases")


if __name__ == "__main__":
    main()
