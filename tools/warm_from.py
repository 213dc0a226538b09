"""Development helper (GPU): the warm solve of tools/trace.py from a labelling in a file, so that two libraries start from
the SAME labelling.  usage: python tools/warm_from.py K N save|load file.npy [pert]"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from phylo_hmrf_amd import Block, synthetic
from phylo_hmrf_amd.tree import PhyloTree
K, S, N, mode, path = int(sys.argv[1]), 4, int(sys.argv[2]), sys.argv[3], sys.argv[4]
pert = float(sys.argv[5]) if len(sys.argv) > 5 else 0.05
tree = PhyloTree(synthetic.tree_for(S)); rng = np.random.default_rng(0)
P = synthetic.sample_ou_params(rng, tree, K); mu, cv = tree.mean_cov(P); cv = cv + 1e-3 * np.eye(S)
P2 = np.clip(P * (1 + 0.15 * rng.standard_normal(P.shape)), 1e-3, 50); mu2, cv2 = tree.mean_cov(P2); cv2 = cv2 + 1e-3 * np.eye(S)
dev = torch.device("cuda", 0)
X = synthetic.device_observations(torch, dev, 1, N, N, True, K, mu, cv); torch.cuda.synchronize()
n = N * (N + 1) // 2
b = Block(n, S, K); b.set_observations_dev(X.data_ptr()); b.sync(); b.build_grid_graph(N, N, True, 8, 0.5)
if mode == "save":
    b.emission(mu2, cv2)
    res = b.solve(1.0, energy_tol_ppb=1000, init_mode=1)
    np.save(path, b.get_labels().astype(np.uint8))
    print("cold solve:", {k: res[k] for k in ("energy", "rounds")})
    sys.exit(0)
lab = np.load(path).astype(np.int32)
P3 = np.clip(P2 * (1 + pert * rng.standard_normal(P.shape)), 1e-3, 50); mu3, cv3 = tree.mean_cov(P3); cv3 = cv3 + 1e-3 * np.eye(S)
for tol in (1000, 0):
    b.set_labels(lab); b.emission(mu3, cv3); b.enable_timing(True); b.reset_timing()
    res = b.solve(1.0, energy_tol_ppb=tol)
    print("warm solve tol %d: energy %.3f (start %.3f) rounds %d changed %d  %s" % (tol, res["energy"], res["energy_init"], res["rounds"], res["changed"],
          {k: round(v[0], 2) for k, v in b.timing().items() if v[1] and k in ("strip", "fusion", "component", "propose")}))
