cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import rvtests_amd, bench
dev=torch.device("cuda:0"); N=500000; V=1024
eng=rvtests_amd.Engine(0); ld=eng.padded_ld(N)
X,y,res,s2=bench.fit_null_qt(dev,N,7)
eng.set_null(0,np.asfortranarray(X.cpu().numpy()),res.cpu().numpy().copy(),np.full(N,float(s2)),float(s2))
blocks,Ms,afs=bench.make_genes(dev,N,ld,1,5,V,V)
hard=torch.round(blocks[0]).contiguous()
bad=((hard!=0)&(hard!=1)&(hard!=2)).sum().item()
print("bad", bad, hard.shape, hard.dtype, hard.max().item(), hard.min().item(), eng.classify_block(hard.data_ptr(),V), eng.classify_block(blocks[0].data_ptr(),V))
dos=blocks[0].clone(); dos += (dos>0)*0.125*torch.rand_like(dos); torch.cuda.synchronize()
eng.cov_block(dos.data_ptr(),V)
print("after dos:", eng.classify_block(hard.data_ptr(),V))
eng.cov_block(hard.data_ptr(),V)
print("after hard:", eng.classify_block(hard.data_ptr(),V))
PY
