#!/usr/bin/env python3
"""Generate tests/golden/davies_liu.json from the COMPILED REFERENCE fragment oracle/_ref/libref_mixchisq.so
(regression/MixtureChiSquare.cpp + qfc.c + cdflib.cpp built where they lie under /root/reference by
`make -C oracle ref`).  Inputs (lambda sets, Q) and the reference's getPvalue / getLiuPvalue outputs, plus the
known-answer cases of regression/test/testMixtureChiSquare.cpp:13-38.  Run only in the build container."""
import json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc

def main():
    assert orc.ref() is not None, "build oracle/_ref first (needs /root/reference)"
    rng = np.random.default_rng(20260101)
    cases = []
    def add(lam, Q, tag=""):
        lam = [float(x) for x in lam]
        cases.append({"lambda": lam, "Q": float(Q), "davies": orc.davies(lam, Q, "ref"), "liu": orc.liu(lam, Q, "ref"),
                      "tag": tag})
    for lam, Q in [([1, 2, 3], 4), ([1, 1, 1], 30), ([1, 1, 1], 50)]:
        add(lam, Q, "testMixtureChiSquare.cpp")
    for t in range(400):
        r = int(rng.integers(2, 81))
        kind = t % 4
        if kind == 0:
            lam = rng.gamma(0.5, 1.0, r)
        elif kind == 1:
            lam = rng.uniform(0.5, 1.5, r) * 10 ** rng.uniform(-3, 6)
        elif kind == 2:
            lam = np.concatenate([[rng.uniform(10, 100)], rng.gamma(1.0, 0.1, r - 1)])
        else:
            lam = 10 ** rng.uniform(-12, 0, r)
        lam = np.sort(lam)[::-1]
        Q = lam.sum() * float(rng.choice([0.01, 0.1, 0.5, 1, 2, 5, 10, 50])) * rng.uniform(0.5, 1.5)
        add(lam, Q)
    for t in range(40):  # negative and zero Q, single lambda
        r = int(rng.integers(2, 40))
        lam = np.sort(rng.gamma(1.0, 1.0, r))[::-1]
        add(lam, -lam.sum() * 10 ** rng.uniform(-8, 3), "negativeQ")
    for t in range(10):
        add([float(rng.gamma(1, 1))], rng.uniform(0.1, 20), "single")
    json.dump({"source": "oracle/_ref (reference MixtureChiSquare.cpp + qfc.c + cdflib.cpp)", "cases": cases},
              open(os.path.join(HERE, "davies_liu.json"), "w"), indent=0)
    print("wrote", len(cases))

if __name__ == "__main__":
    main()
