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
F = ("skat_Q", "skat_p", "skato_Q", "skato_p", "cmc_p", "zeg_p", "cmc_nonref", "n_poly", "status")
ids = list(range(75))
def via_group(partial):
    grp = rvtests_amd.Group([0, 0])
    grp.fit_null(0, X, y)
    got = []
    for g, (G, af) in enumerate(genes):
        grp.submit_gene(ids[g], G, af)
        if partial and g == 40:
            got += grp.collect(cap=25)
    got += grp.collect()
    grp.close()
    return {r.gene_id: r for r in got}
def via_engine():
    eng = rvtests_amd.Engine(0)
    eng.fit_null(0, X, y)
    for g, (G, af) in enumerate(genes):
        eng.submit_gene(ids[g], G, af)
    out = {r.gene_id: r for r in eng.collect()}
    eng.close()
    return out
a = via_engine()
for name, o in (("engine2", via_engine()), ("group", via_group(False)), ("group_partial", via_group(True)), ("group_partial2", via_group(True))):
    for g in range(75):
        for f in F:
            if getattr(a[g], f) != getattr(o[g], f):
                print(name, "gene", g, "masked" if g % 4 == 0 else "hard", "M", genes[g][0].shape[1], f, getattr(a[g], f), getattr(o[g], f))
print("done")
for g in range(75):
    if abs(a[g].skato_p - 0.70186459458639) < 1e-9:
        print("gene", g, "masked" if g % 4 == 0 else "hard", "M", genes[g][0].shape[1], repr(a[g].skato_p), "common" if g % 5 == 1 else "")
