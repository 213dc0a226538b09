"""Development helper (GPU): the cold solve of one block of a workload with and without the coarse-to-fine start.
usage: python tools/c2f_probe.py [workload] [block index] [seed]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from phylo_hmrf_amd import Block, synthetic, workloads
from phylo_hmrf_amd.tree import PhyloTree
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
bi = int(sys.argv[2]) if len(sys.argv) > 2 else 0
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
blocks_def, S, K, nn, desc = workloads.workload(wl)
dev = torch.device("cuda", 0)
tree = PhyloTree(synthetic.tree_for(S)); rng = np.random.default_rng(seed)
P = synthetic.sample_ou_params(rng, tree, K); mu, cv = tree.mean_cov(P); cv = cv + 1e-3 * np.eye(S)
H, W, diag = blocks_def[bi]
n = workloads.block_nodes(H, W, diag); b = Block(n, S, K)
Xd = synthetic.device_observations(torch, dev, bi, H, W, diag, K, mu, cv); torch.cuda.synchronize()
b.set_observations_dev(Xd.data_ptr()); b.sync(); del Xd; b.build_grid_graph(H, W, diag, nn, 0.5)
cur = np.clip(P * (1 + 0.15 * rng.standard_normal(P.shape)), 1e-3, 50)
means, covars = tree.mean_cov(cur); covars = covars + 1e-3 * np.eye(S)
b.emission(means, covars)
tol = int(os.environ.get("TOL_PPB", "1000"))
for rep in range(2):
    for name, kw in (("argmax + ICM start, no coarse-to-fine", dict(pre=True, coarse_start=-1)),
                     ("coarse-to-fine start", dict(pre=False, coarse_start=1))):
        if kw["pre"]:
            b.solve_fast(1.0, max_rounds=1, use_chains=False, use_components=False, use_strips=False, use_expansion=False, init_mode=1)
        b.sync()
        t0 = time.time()
        r = b.solve(1.0, energy_tol_ppb=tol, init_mode=0 if kw["pre"] else 1, coarse_start=kw["coarse_start"])
        b.sync()
        print("%-40s %8.1f ms  rounds %2d  energy %.3f  init %.3f  changed %d" % (name, (time.time() - t0) * 1e3, r["rounds"], r["energy"], r["energy_init"], r["changed"]))
