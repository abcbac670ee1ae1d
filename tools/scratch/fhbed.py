# bed hand-off only, with the submit trace (scratch): RVT_TRACE_SUBMIT=1 python tools/scratch/fhbed.py [reg]
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import rvtests_amd, synth
N = 500000
reg = len(sys.argv) > 1 and sys.argv[1] == "reg"
eng = rvtests_amd.Engine(0)
rng = np.random.default_rng(1)
X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=3)
eng.fit_null(0, X, y)
beds = []
for k in range(4):
    G = rng.binomial(2, 0.01, size=(N, 50)).astype(np.int8)
    beds.append(eng.pack_bed(G))
if reg:
    for b in beds: eng.host_register(b)
for rep in range(2):
    t0 = time.perf_counter()
    n = 512
    for g in range(n):
        eng.submit_gene_bed(g, beds[g % 4], 50, want_af=False)
        if (g + 1) % 64 == 0: eng.collect_ready()
    eng.collect()
    dt = time.perf_counter() - t0
    print("registered" if reg else "pageable", rep, round(n / dt, 1), "genes/s")
eng.close()
