"""Development helper (GPU): the bench's EM loop on a workload, printing per EM iteration the cost, whether labels_local
was renewed, and how far the warm-started solves moved.  usage: python tools/em_diag.py [workload] [iterations] [seed]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from phylo_hmrf_amd import Block, mstep, synthetic, workloads
from phylo_hmrf_amd.base import SLOT_LOCAL
from phylo_hmrf_amd.block import unpack_stats
from phylo_hmrf_amd.tree import PhyloTree
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 25
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
blocks_def, S, K, nn, desc = workloads.workload(wl)
dev = torch.device("cuda", 0)
tree = PhyloTree(synthetic.tree_for(S)); rng = np.random.default_rng(seed)
P = synthetic.sample_ou_params(rng, tree, K); mu, cv = tree.mean_cov(P); cv = cv + 1e-3 * np.eye(S)
blocks = []
for bi, (H, W, diag) in enumerate(blocks_def):
    n = workloads.block_nodes(H, W, diag); b = Block(n, S, K)
    Xd = synthetic.device_observations(torch, dev, seed * 1000 + bi, H, W, diag, K, mu, cv); torch.cuda.synchronize()
    b.set_observations_dev(Xd.data_ptr()); b.sync(); del Xd; b.build_grid_graph(H, W, diag, nn, 0.5); blocks.append(b)
N = sum(b.n for b in blocks)
if os.environ.get("PHMRF_SOLVE_TRACE"):
    for b in blocks: b.enable_timing(True)         # (the trace prints the labels changed per move type with the timers on)
cur = np.clip(P * (1 + 0.15 * rng.standard_normal(P.shape)), 1e-3, 50); init_ou = cur.copy()
means, covars = tree.mean_cov(cur); covars = covars + 1e-3 * np.eye(S)
for b in blocks:
    b.emission(means, covars); b.solve_fast(1.0, max_rounds=1, use_chains=False, use_components=False, use_strips=False, use_expansion=False, init_mode=1)
    b.save_labels(SLOT_LOCAL); b.sync()
ns = K * (1 + S + S * S); min_cost = 1e30
for it in range(iters):
    tot = np.zeros(ns + 4); ch = 0; rounds = 0; t0 = time.time(); took_local = 0; hist = [0] * 10; far = 0; fmax = 0.0
    for b in blocks:
        b.emission(means, covars)
        if os.environ.get("PHMRF_DIAG_WARM", "best") == "best":
            ec, es, took = b.warm_start(1.0, SLOT_LOCAL); took_local += int(took)
        else:
            b.restore_labels(SLOT_LOCAL)
        r = b.solve(1.0, energy_tol_ppb=1000); ch += r["changed"]; rounds += r["rounds"]; hist[min(r["rounds"], 9)] += 1
        far += int(r["changed"] * 8 >= b.n); fmax = max(fmax, r["changed"] / b.n)
        st, costs, _ = b.posterior_stats(1.0, 3)
        tot[:K] += st["post"]; tot[K:K + K * S] += st["obs"].ravel(); tot[K + K * S:ns] += st["obs*obs.T"].ravel(); tot[ns:] += costs
    cost1 = tot[ns + 3] / N; renewed = cost1 < min_cost
    if renewed:
        min_cost = cost1
        for b in blocks: b.save_labels(SLOT_LOCAL)
    p, means, covars, _ = mstep.do_mstep(tree, unpack_stats(tot[:ns], K, S), cur, init_ou, N, 1.0, 0, 0.3, 0.1, 1.0, rng, workers=min(K, os.cpu_count()))
    dpar = float(np.max(np.abs(p - cur) / (np.abs(cur) + 1e-3))); cur = p
    print("it %2d cost1 %.6f %s changed %.2f%% rounds/solve %.1f %s started from labels_local in %d of %d blocks  max rel param change %.3f  blocks that moved >= 1/8: %d (most moved block %.1f%%)  (%.2fs sequential)" % (it, cost1, "renewed" if renewed else "kept   ", 100.0 * ch / N, rounds / len(blocks), hist[1:8], took_local, len(blocks), dpar, far, 100 * fmax, time.time() - t0), flush=True)
mstep.close_pool()
