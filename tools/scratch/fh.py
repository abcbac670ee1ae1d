# from-host rates only (scratch): python tools/scratch/fh.py
import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import bench, rvtests_amd
dev = torch.device("cuda:0")
N = 500000; ld = (N + 15) // 16 * 16
eng = rvtests_amd.Engine(0)
blocks, Ms, afs = bench.make_genes(dev, N, ld, 16, 20260002, 40, 60, 0.0)
X, y = bench.make_phenotype(dev, N, 20260002)
eng.fit_null(0, np.asfortranarray(X.cpu().numpy()), y.cpu().numpy().copy())
out = bench.from_host_rates(eng, blocks, Ms, afs, N)
for k, v in out.items():
    print(k, round(v["gene_sets_per_s"], 1), "genes/s", round(v["host_GBps"], 2), "GB/s")
