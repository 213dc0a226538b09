"""Where the host M-step spends its time (development helper; CPU only).  usage: python tools/mstep_phases.py [workers]"""
import os, sys, time, pickle
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from phylo_hmrf_amd import mstep
from phylo_hmrf_amd.tree import PhyloTree
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
G = os.path.join(ROOT, "tests", "golden")
g1 = np.load(os.path.join(G, "tree_tables.npz")); g = np.load(os.path.join(G, "mstep_objective.npz"))
t = PhyloTree(g1["t4_edge_list"])
K0 = g["t4_post"].shape[0]; K = 20
idx = [i % K0 for i in range(K)]
stats = {"post": g["t4_post"][idx], "obs": g["t4_obs"][idx], "obs*obs.T": g["t4_obsobsT"][idx]}
cur = g["t4_params"][idx]
workers = int(sys.argv[1]) if len(sys.argv) > 1 else 20
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for w in (1, workers):
    for rep in range(5):
        rng = np.random.default_rng(1)
        t0 = time.perf_counter()
        mstep.do_mstep(t, stats, cur, cur, 5000, 1.0, 0, 0.3, 0.1, 1.0, rng, workers=w)
        print("workers %d rep %d: %.2f ms" % (w, rep, (time.perf_counter() - t0) * 1e3))
a = (t, stats["post"][0], stats["obs"][0], stats["obs*obs.T"][0], 5000, 1.0, [cur[0]] * 3, cur[0])
t0 = time.perf_counter(); blob = pickle.dumps(a); t1 = time.perf_counter(); pickle.loads(blob); t2 = time.perf_counter()
print("one task: pickle %.3f ms, unpickle %.3f ms, %d bytes" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, len(blob)))
for c in range(3):
    a = (t, stats["post"][c], stats["obs"][c], stats["obs*obs.T"][c], 5000, 1.0, [cur[c]] * 3, cur[c])
    t0 = time.perf_counter(); mstep._solve_state(a); t1 = time.perf_counter()
    obj = mstep.OUObjective(t, a[1], a[2], a[3], 5000, 1.0); t2 = time.perf_counter()
    x0 = np.clip(cur[c], mstep.LOWER, mstep.UPPER)
    out = mstep._slsqp_native(obj, x0, mstep.LOWER, mstep.UPPER); t3 = time.perf_counter()
    mstep.check_params(t, out[0]); t4 = time.perf_counter(); obj.value(out[0]); t5 = time.perf_counter()
    print("state %d: _solve_state %.2f ms = objective setup %.3f + native SLSQP %.2f + check %.3f + value %.3f"
          % (c, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3))
mstep.close_pool()
