import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from phylo_hmrf_amd import Block, synthetic
from phylo_hmrf_amd.tree import PhyloTree
K,S,N=int(sys.argv[1]),4,int(sys.argv[2])
tree=PhyloTree(synthetic.tree_for(S)); rng=np.random.default_rng(0)
P=synthetic.sample_ou_params(rng,tree,K); mu,cv=tree.mean_cov(P); cv=cv+1e-3*np.eye(S)
P2=np.clip(P*(1+0.15*rng.standard_normal(P.shape)),1e-3,50); mu2,cv2=tree.mean_cov(P2); cv2=cv2+1e-3*np.eye(S)
dev=torch.device("cuda",0)
X=synthetic.device_observations(torch,dev,1,N,N,True,K,mu,cv); torch.cuda.synchronize()
n=N*(N+1)//2
b=Block(n,S,K); b.set_observations_dev(X.data_ptr()); b.sync(); b.build_grid_graph(N,N,True,8,0.5)
b.emission(mu2,cv2)
b.solve_fast(1.0,max_rounds=1,use_chains=False,use_components=False,use_strips=False,use_expansion=False,init_mode=1)
e0=b.energy(1.0)[0]; print("init",e0)
for r in range(40):
    res=b.solve(1.0,max_rounds=1,use_expansion=False)
    print(r,"cheap changed",res["changed"],"E",round(res["energy"],2),"rel",(res["energy"]-e0)/abs(e0))
    if res["changed"]==0: break
res=b.solve(1.0); print("full solve: rounds",res["rounds"],"changed",res["changed"],"E",round(res["energy"],2))
