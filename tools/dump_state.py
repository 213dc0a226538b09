"""GPU box: the inputs of one WARM solve of a synthetic diagonal block, for offline (CPU) probes of the expansions' filter.
usage: python tools/dump_state.py K N out.npz   -> X (f32), the warm solve's unary parameters, the labels it starts from and ends with"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from phylo_hmrf_amd import Block, synthetic
from phylo_hmrf_amd.tree import PhyloTree
K, S, N, out = int(sys.argv[1]), 4, int(sys.argv[2]), sys.argv[3]
tree = PhyloTree(synthetic.tree_for(S)); rng = np.random.default_rng(0)
P = synthetic.sample_ou_params(rng, tree, K); mu, cv = tree.mean_cov(P); cv = cv + 1e-3 * np.eye(S)
P2 = np.clip(P * (1 + 0.15 * rng.standard_normal(P.shape)), 1e-3, 50); mu2, cv2 = tree.mean_cov(P2); cv2 = cv2 + 1e-3 * np.eye(S)
dev = torch.device("cuda", 0)
X = synthetic.device_observations(torch, dev, 1, N, N, True, K, mu, cv); torch.cuda.synchronize()
n = N * (N + 1) // 2
b = Block(n, S, K); b.set_observations_dev(X.data_ptr()); b.sync(); b.build_grid_graph(N, N, True, 8, 0.5)
b.emission(mu2, cv2)
res = b.solve(1.0, energy_tol_ppb=1000, init_mode=1)
print("cold solve:", res)
lab0 = b.get_labels().copy()
P3 = np.clip(P2 * (1 + float(os.environ.get("PHMRF_TRACE_PERT", "0.05")) * rng.standard_normal(P.shape)), 1e-3, 50); mu3, cv3 = tree.mean_cov(P3); cv3 = cv3 + 1e-3 * np.eye(S)
b.emission(mu3, cv3)
lp = b.get_logprob().astype(np.float32)
res = b.solve(1.0, energy_tol_ppb=1000, init_mode=0)
print("warm solve:", res)
lab1 = b.get_labels().copy()
np.savez_compressed(out, X=X.cpu().numpy(), lab0=lab0, lab1=lab1, logprob=lp, N=N, K=K, mu=mu3, cv=cv3)
print("saved", out, "moved", int((lab0 != lab1).sum()), "of", n)
