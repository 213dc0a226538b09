import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from phylo_hmrf_amd import Block, synthetic
from phylo_hmrf_amd.tree import PhyloTree
K,S,N=10,4,2000
tree=PhyloTree(synthetic.tree_for(S)); rng=np.random.default_rng(0)
P=synthetic.sample_ou_params(rng,tree,K); mu,cv=tree.mean_cov(P); cv=cv+1e-3*np.eye(S)
dev=torch.device("cuda",0)
X=synthetic.device_observations(torch,dev,1,N,N,True,K,mu,cv); torch.cuda.synchronize()
n=N*(N+1)//2
b=Block(n,S,K); b.set_observations_dev(X.data_ptr()); b.sync(); b.build_grid_graph(N,N,True,8,0.5)
b.emission(mu,cv); b.solve_fast(1.0,max_rounds=2,use_expansion=False,init_mode=1); b.sync()
b.enable_timing(True)
for name,fn in [("strip_a3",lambda: b.strip_pass(1.0,0,0,0,3)),("strip_fusion",lambda: b.strip_pass(1.0,1,2,5,-1)),("chain0",lambda: b.chain_sweep(1.0,0)),("chain2",lambda: b.chain_sweep(1.0,2)),("icm",lambda: b.icm_sweep(1.0)),("comp",lambda: b.component_pass(1.0))]:
    b.reset_timing(); 
    for _ in range(5): fn()
    t=b.timing(); print(name, {k:(round(v[0]/max(v[1],1)*1e3,1),v[1]) for k,v in t.items() if v[1]})
