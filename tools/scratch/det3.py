import sys, os, subprocess, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
F = ("skat_Q", "skat_p", "skato_Q", "skato_p", "cmc_p", "zeg_p", "cmc_nonref", "n_poly", "status", "zeg_stat", "cmc_stat")
if len(sys.argv) > 1:
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
    for g, (G, af) in enumerate(genes):
        eng.submit_gene(g, G, af)
    out = {r.gene_id: [repr(getattr(r, f)) for f in F] for r in eng.collect()}
    print("RESULT" + json.dumps(out))
    sys.exit(0)
res = {}
for poison in ("0", "255", "127", "85"):
    env = dict(os.environ, RVT_POISON=poison)
    o = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        print("poison", poison, "FAILED", o.stdout[-500:], o.stderr[-1500:])
        continue
    res[poison] = json.loads(line[0][6:])
base = res.get("0")
for poison, r in res.items():
    nd = 0
    for g in base:
        for k, f in enumerate(F):
            if base[g][k] != r[g][k]:
                nd += 1
                if nd < 12:
                    print("poison", poison, "gene", g, f, base[g][k], r[g][k])
    print("poison", poison, "differences:", nd)
