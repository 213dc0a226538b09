"""Development helper (GPU): a FAR-OFF warm start -- the labelling of one parameter set solved under strongly perturbed
parameters -- with and without the coarse scales.  usage: python tools/faroff_probe.py K N mean_run pert [noise]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from phylo_hmrf_amd import Block, synthetic
from phylo_hmrf_amd.tree import PhyloTree
K, S, N, run, pert = int(sys.argv[1]), 4, int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
noise = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
tree = PhyloTree(synthetic.tree_for(S)); rng = np.random.default_rng(0)
P = synthetic.sample_ou_params(rng, tree, K); mu, cv = tree.mean_cov(P); cv = cv + 1e-3 * np.eye(S)
P2 = np.clip(P * (1 + 0.15 * rng.standard_normal(P.shape)), 1e-3, 50); mu2, cv2 = tree.mean_cov(P2); cv2 = cv2 + 1e-3 * np.eye(S)
dev = torch.device("cuda", 0)
X = synthetic.device_observations(torch, dev, 1, N, N, True, K, mu, cv, mean_run=run, noise=noise); torch.cuda.synchronize()
n = N * (N + 1) // 2
b = Block(n, S, K); b.set_observations_dev(X.data_ptr()); b.sync(); b.build_grid_graph(N, N, True, 8, 0.5)
b.enable_timing(True)
b.emission(mu2, cv2)
b.solve_fast(1.0, max_rounds=1, use_chains=False, use_components=False, use_strips=False, use_expansion=False, init_mode=1)
t0 = time.time(); r = b.solve(1.0, energy_tol_ppb=1000); b.sync()
print("cold (argmax + ICM start): %.1f ms rounds %d energy %.2f changed %.1f %%" % ((time.time() - t0) * 1e3, r["rounds"], r["energy"], 100.0 * r["changed"] / n))
b.save_labels(1)
P3 = np.clip(P2 * (1 + pert * rng.standard_normal(P.shape)), 1e-3, 50); mu3, cv3 = tree.mean_cov(P3); cv3 = cv3 + 1e-3 * np.eye(S)
b.emission(mu3, cv3)
for name, kw in (("coarse scales on", dict()), ("coarse scales off", dict(use_coarse=False)), ("coarse scales on", dict()), ("coarse scales off", dict(use_coarse=False))):
    b.restore_labels(1); b.sync(); b.reset_timing()
    t0 = time.time(); r = b.solve(1.0, energy_tol_ppb=1000, **kw); b.sync()
    print("far-off warm start, %-17s %7.1f ms rounds %2d energy %.2f (start %.2f) changed %.1f %%  %s" % (
        name + ":", (time.time() - t0) * 1e3, r["rounds"], r["energy"], r["energy_init"], 100.0 * r["changed"] / n,
        {k: round(v[0], 1) for k, v in b.timing().items() if v[1]}))
