# group hand-off with 1 / 2 members on one GPU, submit trace per member (scratch)
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import rvtests_amd, synth
N, M = 500000, 50
rng = np.random.default_rng(1)
X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=3)
e0 = rvtests_amd.Engine(0)
beds = [e0.pack_bed(rng.binomial(2, 0.01, size=(N, M)).astype(np.int8)) for _ in range(4)]
e0.close()
for members in (1, 2):
    grp = rvtests_amd.Group([0] * members)
    grp.fit_null(0, X, y)
    for rep in range(2):
        t0 = time.perf_counter()
        n = 2048
        for g in range(n):
            grp.submit_gene_bed(g, beds[g % 4], M)
            if (g + 1) % 64 == 0: grp.collect_ready()
        grp.collect()
        print("members", members, "rep", rep, round(n / (time.perf_counter() - t0), 1), "genes/s")
    grp.close()
