import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, rvtests_amd, bench
dev=torch.device('cuda',0)
N=500000
eng=rvtests_amd.Engine(0); ld=eng.padded_ld(N)
X,y,res,s2=bench.fit_null_qt(dev,N,20260002)
Xh=np.asfortranarray(X.cpu().numpy()); resh=res.cpu().numpy().copy()
eng.set_null(0,Xh,resh,np.full(N,s2),s2)
blocks,Ms,afs=bench.make_genes(dev,N,ld,32,20260002,20,80)
torch.cuda.synchronize()
out=eng.run_blocks([b.data_ptr() for b in blocks],Ms,afs)
out=eng.run_blocks([b.data_ptr() for b in blocks],Ms,afs)
tot=np.array([r.zeg_U for r in out]); dv=np.array([r.cmc_U for r in out]); li=np.array([r.cmc_V for r in out]); ne=np.array([r.zeg_V for r in out]); tm=np.array([r.davies_terms for r in out])
print("per gene: total Mcycles %.1f  davies %.1f  liu+density %.1f  rest %.1f | neval %.0f terms %.0f"%(tot.mean()/1e6,dv.mean()/1e6,li.mean()/1e6,(tot-dv-li).mean()/1e6,ne.mean(),tm.mean()))
for r,M in list(zip(out,Ms))[:8]: print(M, r.n_poly, "tot %.1f dav %.1f liu %.1f neval %d terms %d p %.3g"%(r.zeg_U/1e6,r.cmc_U/1e6,r.cmc_V/1e6,r.zeg_V,r.davies_terms,r.skato_p))
