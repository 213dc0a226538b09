"""Development helper (GPU): one cold solve (the first E-step of the bench) of one block of a workload with the per-round
trace of the solver (PHMRF_SOLVE_TRACE=1): rounds, labels changed, energy, ms per kernel class.
usage: PHMRF_SOLVE_TRACE=1 python tools/cold_trace.py [workload] [block index] [seed]"""
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
b.enable_timing(True)
b.emission(means, covars)
b.solve_fast(1.0, max_rounds=1, use_chains=False, use_components=False, use_strips=False, use_expansion=False, init_mode=1)
b.sync(); b.reset_timing()
t0 = time.time()
r = b.solve(1.0, energy_tol_ppb=int(os.environ.get("TOL_PPB", "1000")))
b.sync()
print("block %d of %s: %d nodes, cold solve %.1f ms, %s" % (bi, wl, n, (time.time() - t0) * 1e3, {k: r[k] for k in ("rounds", "changed", "energy", "converged") if k in r}))
print({k: (round(v[0], 1), v[1]) for k, v in b.timing().items()})
if os.environ.get("COLD_TWICE"):          # the same cold solve again: the coarse child blocks exist now (allocated by the first)
    b.solve_fast(1.0, max_rounds=1, use_chains=False, use_components=False, use_strips=False, use_expansion=False, init_mode=1)
    b.sync(); b.reset_timing()
    t0 = time.time()
    r = b.solve(1.0, energy_tol_ppb=int(os.environ.get("TOL_PPB", "1000")))
    b.sync()
    print("again: cold solve %.1f ms, rounds %d" % ((time.time() - t0) * 1e3, r["rounds"]))
import hashlib
print("labels sha1 %s energy %.9f rounds %d" % (hashlib.sha1(b.get_labels().tobytes()).hexdigest()[:16], r["energy"], r["rounds"]))
