"""Host-side wall time of the calls of one block's E-step (development helper, GPU only): where a single block's
E-step spends its time outside the kernels.   usage: python tools/estep_phases.py K N [iterations] [perturbation]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from phylo_hmrf_amd import Block, synthetic
from phylo_hmrf_amd.tree import PhyloTree
K, S, N = int(sys.argv[1]), int(os.environ.get("PHMRF_TOOL_S", "4")), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 8
pert = float(sys.argv[4]) if len(sys.argv) > 4 else 0.005
tree = PhyloTree(synthetic.tree_for(S)); rng = np.random.default_rng(0)
P = synthetic.sample_ou_params(rng, tree, K); mu, cv = tree.mean_cov(P); cv = cv + 1e-3 * np.eye(S)
dev = torch.device("cuda", 0)
X = synthetic.device_observations(torch, dev, 1, N, N, True, K, mu, cv); torch.cuda.synchronize()
n = N * (N + 1) // 2
b = Block(n, S, K); b.set_observations_dev(X.data_ptr()); b.sync(); b.build_grid_graph(N, N, True, 8, 0.5)
P2 = np.clip(P * (1 + 0.15 * rng.standard_normal(P.shape)), 1e-3, 50); mu2, cv2 = tree.mean_cov(P2); cv2 = cv2 + 1e-3 * np.eye(S)
b.emission(mu2, cv2)
b.solve(1.0, energy_tol_ppb=1000, init_mode=1)
b.save_labels(0)
for timing in (False, True):
    b.enable_timing(timing)
    rows = []
    for it in range(iters):
        P2 = np.clip(P2 * (1 + pert * rng.standard_normal(P.shape)), 1e-3, 50); mu3, cv3 = tree.mean_cov(P2); cv3 = cv3 + 1e-3 * np.eye(S)
        b.sync(); b.reset_timing() if timing else None
        t = [time.perf_counter()]
        b.restore_labels(0); t.append(time.perf_counter())
        b.emission(mu3, cv3); t.append(time.perf_counter())
        res = b.solve_fast(1.0, energy_tol_ppb=1000); t.append(time.perf_counter())
        stats, costs, _ = b.posterior_stats(1.0, 3); t.append(time.perf_counter())
        b.save_labels(0); b.sync(); t.append(time.perf_counter())
        d = np.diff(t) * 1e3
        extra = ""
        if timing:
            tm = b.timing()
            extra = " kernels %.2f ms: %s" % (sum(v[0] for v in tm.values()), {k: (round(v[0], 2), v[1]) for k, v in tm.items() if v[1]})
        rows.append(d)
        print("timing=%d it %d: restore %.2f emission %.2f solve %.2f posterior %.2f save %.2f  total %.2f ms  rounds %s%s"
              % (timing, it, d[0], d[1], d[2], d[3], d[4], d.sum(), "-", extra))
    print("mean of last half:", np.mean(rows[len(rows) // 2:], axis=0).round(2), "sum", np.mean(rows[len(rows) // 2:], axis=0).sum().round(2))
