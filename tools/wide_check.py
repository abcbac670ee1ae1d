import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, rvtests_amd, orc, synth
eng = rvtests_amd.Engine(0)
N = 3000
X, y, res, v, s2 = synth.make_null(N, 3, 0, seed=1)
eng.set_null(0, X, res, v, s2)
for M in (400, 1000):
    _, G, af = synth.make_gene(N, M, seed=M, missing=0.005, common=True, mono=True)
    ptr = eng.upload_block(G)
    t0 = time.time(); r = eng.run_blocks([ptr], [M], [af], tests=rvtests_amd.TEST_SKAT | 12)[0]; t1 = time.time()
    rc, a = orc.skat(G, af, X, res, v, 0); t2 = time.time()
    print(M, r.n_poly, a.n_poly, r.skat_Q, a.Q, r.skat_p, a.pvalue, "gpu %.2fs cpu %.2fs" % (t1 - t0, t2 - t1))
    rc, c = orc.burden(G, X, y, 0, 0)
    print("  cmc", r.cmc_nonref, c.nonref_site, r.cmc_p, c.pvalue)
