#!/usr/bin/env python3
"""Generate tests/golden/kbac.json from the REFERENCE's own KBAC code: regression/kbac.cpp + kbac_interface.cpp compiled
where they lie and linked against GSL 1.16 built from the reference's vendored tarball (third/gsl-1.16.tar.gz; see
make_gsl_golden.py for the build recipe, prefix /tmp/gslinst; its headers are also reachable through the path the
reference includes them by, third/gsl/include/gsl, under GSL_INCROOT = /tmp/gslinc).  Each case: a small genotype matrix
(values 0/1/2 and a few imputed non-integers), a 0/1 phenotype, the per-column frequencies, nPerm and alpha as
KBACTest::fit passes them (src/Model.h:2925-2998: quiet = 1, mafUpper = 1, sided = 1), srand(seed) first.  Recorded: the
p-value and the NEXT rand() value (pins how far the shuffles advanced the process-wide stream).  Also a table of
gsl_cdf_hypergeometric_P values.  Run ONLY in the build container."""
import json
import os
import subprocess
import tempfile

import numpy as np

REF = "/root/reference"
GSL = os.environ.get("GSL_PREFIX", "/tmp/gslinst")
INCROOT = os.environ.get("GSL_INCROOT", "/tmp/gslinc")
HERE = os.path.dirname(os.path.abspath(__file__))

DRIVER = r'''
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kbac_interface.h"
#include "third/gsl/include/gsl/gsl_cdf.h"
int main() {
  char op[16];
  while (scanf("%15s", op) == 1) {
    if (op[0] == 'k') {
      int N, M, nperm; unsigned seed; double alpha;
      if (scanf("%d %d %d %lf %u", &N, &M, &nperm, &alpha, &seed) != 5) return 1;
      std::vector<double> x((size_t)N * M), y(N), maf(M);
      for (auto& v : x) if (scanf("%lf", &v) != 1) return 1;     // people-major: person 1 marker 1..M, person 2 ...
      for (auto& v : y) if (scanf("%lf", &v) != 1) return 1;
      for (auto& v : maf) if (scanf("%lf", &v) != 1) return 1;
      int nn = nperm, qq = 1, xcol = M, ylen = N, twosided = 1;
      double aa = alpha, mafUpper = 1.0, p = 9.0;
      srand(seed);
      set_up_kbac_test(&nn, &qq, &aa, &mafUpper, x.data(), y.data(), maf.data(), &xcol, &ylen);
      do_kbac_test(&p, &twosided);
      clear_kbac_test();
      printf("RESK %.17g %d\n", p, rand());
    } else {
      unsigned k, n1, n2, t;
      if (scanf("%u %u %u %u", &k, &n1, &n2, &t) != 4) return 1;
      printf("RESH %.17g\n", gsl_cdf_hypergeometric_P(k, n1, n2, t));
    }
    fflush(stdout);
  }
  return 0;
}
'''


def main():
    rng = np.random.default_rng(20260002)
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "drv.cpp")
        open(src, "w").write(DRIVER)
        exe = os.path.join(td, "drv")
        subprocess.check_call(["g++", "-O2", "-w", "-std=c++11", "-I" + os.path.join(REF, "regression"), "-I" + REF,
                               "-I" + INCROOT, "-I" + os.path.join(GSL, "include"), "-o", exe, src, os.path.join(REF, "regression", "kbac.cpp"),
                               os.path.join(REF, "regression", "kbac_interface.cpp"),
                               os.path.join(GSL, "lib", "libgsl.a"), os.path.join(GSL, "lib", "libgslcblas.a"), "-lm"])
        cases, lines = [], []
        shapes = [(40, 3, 300, 0.05), (120, 8, 500, 0.05), (300, 12, 400, 1.0), (257, 40, 300, 0.05),
                  (90, 5, 6000, 0.01), (500, 20, 200, 0.05), (64, 2, 100, 0.05), (150, 60, 150, 0.05)]
        for ci, (N, M, nperm, alpha) in enumerate(shapes):
            maf = rng.uniform(0.002, 0.08, M)
            if M > 4:
                maf[2] = 0.0                                   # trimmed column (maf <= mafLower)
            G = (rng.random((N, M)) < 2 * maf[None, :] + 0.01).astype(float)
            G[rng.random((N, M)) < 0.004] = 2.0
            if ci % 2 == 1:
                G[rng.integers(N), rng.integers(M)] = 0.37    # an imputed value: "invalid coding" -> wild type
            y = (rng.random(N) < 0.45).astype(float)
            if ci == 4:                                        # a real signal: the adaptive rule does not stop early
                y = ((G.sum(1) > 0) & (rng.random(N) < 0.9) | (rng.random(N) < 0.2)).astype(float)
            seed = 1 + 7 * ci
            lines.append("k %d %d %d %.17g %u\n%s\n%s\n%s\n" % (
                N, M, nperm, alpha, seed, " ".join("%.17g" % v for v in G.ravel()),
                " ".join("%g" % v for v in y), " ".join("%.17g" % v for v in maf)))
            cases.append({"N": N, "M": M, "nperm": nperm, "alpha": alpha, "seed": seed, "G": G.tolist(),
                          "y": y.tolist(), "maf": maf.tolist()})
        hyper = []
        for _ in range(400):
            n1 = int(rng.integers(1, 60))
            n2 = int(rng.choice([50, 300, 5000, 400000]))
            t = int(rng.integers(1, n1 + n2))
            k = int(rng.integers(0, n1 + 1))
            hyper.append([k, n1, n2, t])
            lines.append("h %d %d %d %d\n" % (k, n1, n2, t))
        out = subprocess.run([exe], input="".join(lines), capture_output=True, text=True, check=True).stdout.split("\n")
        resk = [ln.split()[1:] for ln in out if ln.startswith("RESK ")]   # (the reference prints its warnings to stdout)
        resh = [ln.split()[1] for ln in out if ln.startswith("RESH ")]
        assert len(resk) == len(cases) and len(resh) == len(hyper)
        for c, (p, nxt) in zip(cases, resk):
            c["pvalue"] = float(p)
            c["next_rand"] = int(nxt)
        hv = [float(v) for v in resh]
    json.dump({"source": "regression/kbac.cpp + kbac_interface.cpp of the reference, GSL 1.16 from its vendored tarball",
               "cases": cases, "hypergeometric_P": [h + [v] for h, v in zip(hyper, hv)]},
              open(os.path.join(HERE, "kbac.json"), "w"))
    print("wrote kbac.json:", [(c["N"], c["M"], c["pvalue"]) for c in cases])


if __name__ == "__main__":
    main()
