"""Development helper (GPU only): save the solver's labellings of the cfg2 block for offline analysis."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import ref_numpy as R, synth
from phylo_hmrf_amd import Block
seed, N, K = 13, 2000, 10
blk = synth.make_block(seed, N, N, 4, K, True)
X = blk["X"]; n = X.shape[0]
w, eid = R.edge_weights_from_distance(blk["edges"], 0.5)
lp = R.log_multivariate_normal_density_full(X, blk["means"], blk["covars"])
init = np.random.default_rng(seed + 7).integers(0, K, n)
b = Block(n, 4, K); b.set_graph(eid, w); b.set_grid(N, N, True, 8); b.set_logprob(lp)
b.solve(1.0, energy_tol_ppb=0, init_mode=1)
La = b.get_labels()
b.set_labels(init); b.solve(1.0, energy_tol_ppb=0)
Lr = b.get_labels()
np.savez_compressed("gpurun_out/exp_labels.npz", L_argmax=np.uint8(La), L_random=np.uint8(Lr))
