"""Per-round trace of one solve on a synthetic diagonal block (development helper, GPU only).
usage: PHMRF_SOLVE_TRACE=1 python tools/trace.py K N [tol_ppb]"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from phylo_hmrf_amd import Block, synthetic
from phylo_hmrf_amd.tree import PhyloTree
K, S, N = int(sys.argv[1]), 4, int(sys.argv[2])
tol = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
expn = int(sys.argv[4]) if len(sys.argv) > 4 else 1
comp = int(sys.argv[5]) if len(sys.argv) > 5 else 1          # component moves in the WARM solve
tree = PhyloTree(synthetic.tree_for(S)); rng = np.random.default_rng(0)
P = synthetic.sample_ou_params(rng, tree, K); mu, cv = tree.mean_cov(P); cv = cv + 1e-3 * np.eye(S)
P2 = np.clip(P * (1 + 0.15 * rng.standard_normal(P.shape)), 1e-3, 50); mu2, cv2 = tree.mean_cov(P2); cv2 = cv2 + 1e-3 * np.eye(S)
dev = torch.device("cuda", 0)
X = synthetic.device_observations(torch, dev, 1, N, N, True, K, mu, cv); torch.cuda.synchronize()
n = N * (N + 1) // 2
b = Block(n, S, K); b.set_observations_dev(X.data_ptr()); b.sync(); b.build_grid_graph(N, N, True, 8, 0.5)
b.enable_timing(True)
b.emission(mu2, cv2)
res = b.solve(1.0, energy_tol_ppb=tol, init_mode=1, use_expansion=expn)
print("cold solve:", res)
# warm start as in EM iteration >= 1: perturb the parameters slightly, keep labels
P3 = np.clip(P2 * (1 + float(os.environ.get("PHMRF_TRACE_PERT", "0.02")) * rng.standard_normal(P.shape)), 1e-3, 50); mu3, cv3 = tree.mean_cov(P3); cv3 = cv3 + 1e-3 * np.eye(S)
b.emission(mu3, cv3)
sys.stderr.write("---- warm\n")
b.reset_timing()
res = b.solve(1.0, energy_tol_ppb=tol, init_mode=0, use_expansion=expn, use_components=bool(comp))
print("warm solve:", res)
stats, costs, _ = b.posterior_stats(1.0, 3)
print("costs/n:", (costs / n).round(5))
print("work:", b.work())
print("timing:", {k: (round(v[0], 3), v[1]) for k, v in b.timing().items() if v[1]})
