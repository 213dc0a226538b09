"""Development helper (GPU): six block life cycles of 60 warm E-steps each; prints the free device memory after each
(no growth after the first = no leak across phmrf_block_create / close)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from phylo_hmrf_amd import Block, synthetic
from phylo_hmrf_amd.tree import PhyloTree
K, S, N = 20, 4, 1200
tree = PhyloTree(synthetic.tree_for(S)); rng = np.random.default_rng(0)
P = synthetic.sample_ou_params(rng, tree, K); mu, cv = tree.mean_cov(P); cv = cv + 1e-3 * np.eye(S)
dev = torch.device("cuda", 0)
X = synthetic.device_observations(torch, dev, 1, N, N, True, K, mu, cv); torch.cuda.synchronize()
n = N * (N + 1) // 2
free0 = None
for rep in range(6):
    b = Block(n, S, K); b.set_observations_dev(X.data_ptr()); b.sync(); b.build_grid_graph(N, N, True, 8, 0.5)
    P2 = P.copy()
    b.emission(mu, cv); b.solve(1.0, energy_tol_ppb=1000, init_mode=1); b.save_labels(0)
    for it in range(60):
        P2 = np.clip(P2 * (1 + 0.01 * rng.standard_normal(P.shape)), 1e-3, 50); m2, c2 = tree.mean_cov(P2)
        b.restore_labels(0); b.emission(m2, c2 + 1e-3 * np.eye(S)); b.solve_fast(1.0, energy_tol_ppb=1000)
        st, costs, _ = b.posterior_stats(1.0, 3); b.save_labels(0)
        if it % 2 == 0: b.prepare_components()          # (every other E-step with the components prepared ahead)
    b.close(); torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if free0 is None: free0 = free
    print("rep", rep, "free MB", free >> 20, "delta", (free - free0) >> 20, "cost", costs[3] / n)
