# VCF text / BGEN hand-off only (scratch): python tools/scratch/fhvcf.py [bgen]
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import rvtests_amd, synth
N = 500000
bgen = len(sys.argv) > 1 and sys.argv[1] == "bgen"
eng = rvtests_amd.Engine(0)
rng = np.random.default_rng(1)
X, y, res, v, s2 = synth.make_null(N, 2, 0, seed=3)
eng.fit_null(0, X, y)
eng.vcf_set_samples(np.arange(N, dtype=np.int32))
data = []
for k in range(3):
    h = rng.binomial(2, 0.01, size=(N, 50))
    if bgen:
        head = np.array([N], dtype="<u4").tobytes() + np.array([2], dtype="<u2").tobytes() + bytes([2, 2])
        pm = np.full(N, 2, dtype=np.uint8).tobytes() + bytes([0, 16])
        blks = []
        for j in range(50):
            vv = np.zeros((N, 2), dtype="<u2")
            vv[h[:, j] == 0, 0] = 65535
            vv[h[:, j] == 1, 1] = 65535
            blks.append(head + pm + vv.tobytes())
        data.append(blks)
    else:
        lut = np.frombuffer(b"0/0\t0/1\t1/1\t", dtype=np.uint8).reshape(3, 4)
        hd = b"1\t1000\t.\tA\tG\t50\tPASS\t.\tGT\t"
        data.append(eng.prepare_vcf([hd + lut[h[:, j]].tobytes()[:-1] for j in range(50)]))
for rep in range(2):
    t0 = time.perf_counter()
    n = 128
    for g in range(n):
        if bgen:
            eng.submit_gene_bgen(g, data[g % 3], 2, want_af=False)
        else:
            eng.submit_gene_vcf(g, data[g % 3], want_af=False)
        if (g + 1) % 64 == 0: eng.collect_ready()
    eng.collect()
    dt = time.perf_counter() - t0
    print("bgen" if bgen else "vcf", rep, round(n / dt, 1), "genes/s")
eng.close()
