import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, rvtests_amd, synth
N, d = 3001, 3
rng = np.random.default_rng(3)
genes = []
for g in range(75):
    M = int(rng.integers(1, 70))
    Graw, G, af = synth.make_gene(N, M, seed=7000 + g, missing=0.01 if g % 4 == 0 else 0.0, common=(g % 5 == 1))
    genes.append((G, af))
X, y, res, v, s2 = synth.make_null(N, d, 0, seed=9, G_effect=0.4 * genes[3][0][:, :2].sum(1))
eng = rvtests_amd.Engine(0)
eng.fit_null(0, X, y)
F = ("skat_Q", "skat_p", "skato_Q", "skato_p", "cmc_p", "zeg_p", "cmc_nonref", "n_poly", "status")
def run(order_chunks):
    out = {}
    for chunk in order_chunks:
        for g in chunk:
            eng.submit_gene(g, genes[g][0], genes[g][1])
        for r in eng.collect():
            out[r.gene_id] = r
    return out
a = run([list(range(75))])
b = run([list(range(75))])
c = run([list(range(s, min(s + 9, 75))) for s in range(0, 75, 9)])
for name, o in (("repeat", b), ("chunks9", c)):
    for g in range(75):
        for f in F:
            if getattr(a[g], f) != getattr(o[g], f):
                print(name, "gene", g, "masked" if g % 4 == 0 else "hard", "M", genes[g][0].shape[1], f, getattr(a[g], f), getattr(o[g], f))
print("done")
