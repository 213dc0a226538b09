"""Development helper (GPU): one solve of a k-NN graph case of tests/golden/make_golden_knn_gco.py from random labels, with
its wall time, and -- with PHMRF_SOLVE_TRACE=1 -- a line per round.  usage: python tools/knn_trace.py CASE [tol_ppb]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden"))
import numpy as np
import make_golden_knn_gco as G
from oracle import ref_numpy as R
from phylo_hmrf_amd import Block
case = G.CASES[int(sys.argv[1])]
tol = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n, eid, w, lp, init = G.case_inputs(*case)
K = lp.shape[1]
b = Block(n, 4, K); b.set_graph(eid, w); b.set_logprob(lp)
for rep in range(3):                      # (the first solve builds the path families and the arcs' reverse slots on the host)
    b.set_labels(init); b.sync()
    t0 = time.time()
    res = b.solve(1.0, energy_tol_ppb=tol)
    dt = time.time() - t0
    print("solve %d: %.1f ms, rounds %d, energy %.3f (float64 oracle: %.3f)" % (rep, dt * 1e3, res["rounds"], res["energy"],
                                                                               R.mrf_energy(b.get_labels(), lp, eid, w, 1.0)[0]))
