import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden"))
import numpy as np
import make_golden_knn_gco as G
from oracle import ref_numpy as R
from phylo_hmrf_amd import Block
case = G.CASES[int(sys.argv[1])]
n, eid, w, lp, init = G.case_inputs(*case)
K = lp.shape[1]
b = Block(n, 4, K); b.set_graph(eid, w); b.set_logprob(lp); b.set_labels(init)
b.enable_timing(True)
res = b.solve(1.0, energy_tol_ppb=0)
print(res, R.mrf_energy(b.get_labels(), lp, eid, w, 1.0)[0])
print(b.timing())
