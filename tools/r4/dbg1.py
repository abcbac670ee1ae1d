import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, rvtests_amd, synth
eng = rvtests_amd.Engine(0)
N, d = 3000, 2
rng = np.random.default_rng(1)
for M in (1, 7, 30):
    maf = 10 ** rng.uniform(-2.3, -1.0, M)
    G = np.asfortranarray(rng.binomial(2, maf, size=(N, M)).astype(np.float64))
    X, y, res, v, s2 = synth.make_null(N, d, 1, seed=12)
    eng.set_null(1, X, res, v, s2)
    ptr = eng.upload_block(G)
    out = {}
    for hard in (True, False):
        eng.set_hardcall(hard)
        out[hard] = eng.debug_suffstat(ptr, M)
    eng.set_hardcall(True)
    S0, T0, u0 = out[True][:3]
    S1, T1, u1 = out[False][:3]
    Sx = (G * v[:, None]).T @ G
    Tx = (G * v[:, None]).T @ X
    print("M", M, "S hcx vs gen", np.max(np.abs(S0 - S1)), "S hcx vs numpy", np.max(np.abs(S0 - Sx)), "gen vs numpy", np.max(np.abs(S1 - Sx)))
    print("   T hcx vs gen", np.max(np.abs(T0 - T1)), "T hcx vs numpy", np.max(np.abs(T0 - Tx)), " u", np.max(np.abs(u0 - u1)))
    if M == 1:
        print(S0, S1, Sx, T0, T1, Tx)
