#!/usr/bin/env python3
"""Generate tests/golden/gsl_scalar.json from GSL 1.16 — the library the reference vendors as
third/gsl-1.16.tar.gz and links for gsl_ran_beta_pdf / gsl_cdf_chisq_{P,Q,Qinv} / gsl_ran_chisq_pdf /
gsl_integration_qags (src/Model.h:2651, regression/SkatO.cpp:236-256,325,336,421,430,
regression/LinearRegressionScoreTest.cpp:259).

Run ONLY in the build container (needs /root/reference):
    mkdir -p /tmp/gslsrc && tar xzf /root/reference/third/gsl-1.16.tar.gz -C /tmp/gslsrc
    (cd /tmp/gslsrc/gsl-1.16 && ./configure --prefix=/tmp/gslinst --disable-shared && make -j6 && make install)
    python tests/golden/make_gsl_golden.py
The fixture (inputs + GSL outputs) is data; no GSL source is stored in this repository."""
import json, os, subprocess, sys, tempfile
import numpy as np

GSL = os.environ.get("GSL_PREFIX", "/tmp/gslinst")
HERE = os.path.dirname(os.path.abspath(__file__))

C_SRC = r'''
#include <stdio.h>
#include <math.h>
#include <gsl/gsl_cdf.h>
#include <gsl/gsl_randist.h>
#include <gsl/gsl_integration.h>
#include <gsl/gsl_errno.h>
static double f(double x, void* p){ double* q=(double*)p; int id=(int)q[0]; double al=q[1];
  switch(id){
   case 0: return pow(x,al)*log(1/x);
   case 1: return exp(-x)*sin(al*x);
   case 2: return gsl_ran_chisq_pdf(x,1.0)*exp(-al*x);
   case 3: return 1.0/(1.0+al*x*x);
   case 4: return (x>0?pow(x,-0.5):0.0)*cos(al*x);
  } return 0; }
int main(int argc,char**argv){
  gsl_set_error_handler_off();
  char op[32]; double a,b,c,d,e,g; int lim;
  while (scanf("%31s",op)==1){
    if(!strcmp(op,"beta")){ scanf("%lf %lf %lf",&a,&b,&c); printf("%.17g\n",gsl_ran_beta_pdf(a,b,c)); }
    else if(!strcmp(op,"chisqQ")){ scanf("%lf %lf",&a,&b); printf("%.17g\n",gsl_cdf_chisq_Q(a,b)); }
    else if(!strcmp(op,"chisqP")){ scanf("%lf %lf",&a,&b); printf("%.17g\n",gsl_cdf_chisq_P(a,b)); }
    else if(!strcmp(op,"chisqQinv")){ scanf("%lf %lf",&a,&b); printf("%.17g\n",gsl_cdf_chisq_Qinv(a,b)); }
    else if(!strcmp(op,"chisqpdf")){ scanf("%lf %lf",&a,&b); printf("%.17g\n",gsl_ran_chisq_pdf(a,b)); }
    else if(!strcmp(op,"qags")){ int id; scanf("%d %lf %lf %lf %lf %lf %d",&id,&a,&b,&c,&d,&e,&lim);
       double q[2]={(double)id,a}; gsl_function F; F.function=f; F.params=q;
       gsl_integration_workspace* w=gsl_integration_workspace_alloc(lim);
       double res,err; int st=gsl_integration_qags(&F,b,c,d,e,lim,w,&res,&err);
       printf("%d %.17g %.17g %zu\n",st,res,err,w->size); gsl_integration_workspace_free(w); }
  }
  return 0; }
'''

def main():
    rng = np.random.default_rng(20260101)
    cases = []
    for m in np.concatenate([10 ** rng.uniform(-7, -0.302, 150), [1e-30, 0.5, 0.25, 1e-3]]):
        for (b1, b2) in [(1.0, 25.0), (0.5, 0.5), (1.0, 1.0)]:
            cases.append(("beta", [float(m), b1, b2]))
    for _ in range(400):
        df = float(rng.choice([1.0, 0.5 + rng.uniform(0, 3), rng.uniform(1, 60), rng.uniform(60, 400)]))
        x = float(df * 10 ** rng.uniform(-3, 1.3))
        cases.append(("chisqQ", [x, df]))
        cases.append(("chisqP", [x, df]))
    for _ in range(300):
        df = float(rng.choice([1.0, rng.uniform(0.6, 5), rng.uniform(1, 80)]))
        q = float(rng.choice([10 ** rng.uniform(-14, -1.4), rng.uniform(0.05, 0.95), 1 - 10 ** rng.uniform(-6, -1.4)]))
        cases.append(("chisqQinv", [q, df]))
    for x in 10 ** rng.uniform(-6, 1.7, 100):
        cases.append(("chisqpdf", [float(x), 1.0]))
    qags = [(0, 2.6, 0, 1, 0, 1e-10), (0, -0.9, 0, 1, 0, 1e-10), (1, 10.0, 0, 40, 1e-25, 0.0001220703),
            (2, 0.0, 0, 40, 1e-25, 0.0001220703), (2, 0.7, 0, 40, 1e-25, 0.0001220703),
            (2, 3.0, 0, 40, 1e-25, 0.0001220703), (3, 100.0, 0, 40, 1e-25, 0.0001220703),
            (3, 1e4, -1, 1, 0, 1e-8), (4, 5.0, 0, 40, 1e-25, 0.0001220703), (4, 40.0, 0, 10, 0, 1e-9),
            (2, 0.1, 0, 40, 0, 1e-12)]
    for (i, al, a, b, ea, er) in qags:
        cases.append(("qags", [i, al, a, b, ea, er, 1000]))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "g.c"); exe = os.path.join(td, "g")
        open(src, "w").write("#include <string.h>\n" + C_SRC)
        subprocess.check_call(["gcc", "-O2", "-I" + GSL + "/include", src, "-o", exe, "-L" + GSL + "/lib", "-lgsl", "-lgslcblas", "-lm"])
        inp = "\n".join(op + " " + " ".join(repr(v) for v in args) for op, args in cases) + "\n"
        out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert len(out) == len(cases)
    rec = []
    for (op, args), line in zip(cases, out):
        if op == "qags":
            st, res, err, size = line.split()
            rec.append({"op": op, "args": args, "status": int(st), "result": float(res), "abserr": float(err), "intervals": int(size)})
        else:
            rec.append({"op": op, "args": args, "value": float(line)})
    json.dump({"source": "GSL 1.16 built from /root/reference/third/gsl-1.16.tar.gz", "cases": rec},
              open(os.path.join(HERE, "gsl_scalar.json"), "w"), indent=0)
    print("wrote", len(rec), "cases")

if __name__ == "__main__":
    main()
