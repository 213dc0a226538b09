"""Development helper (GPU only): final energy of the label solver on the cfg2-size synthetic block from different starts.
usage: python tools/exp_energy.py [seed N K]   (gco reference energies for seed 13, N 2000, K 10 are known constants)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import ref_numpy as R, synth
from phylo_hmrf_amd import Block
seed, N, K = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (13, 2000, 10)
blk = synth.make_block(seed, N, N, 4, K, True)
X = blk["X"]; n = X.shape[0]
w, eid = R.edge_weights_from_distance(blk["edges"], 0.5)
lp = R.log_multivariate_normal_density_full(X, blk["means"], blk["covars"])
init = np.random.default_rng(seed + 7).integers(0, K, n)
b = Block(n, 4, K); b.set_graph(eid, w); b.set_grid(N, N, True, 8); b.set_logprob(lp)
E = lambda lab: R.mrf_energy(lab, lp, eid, w, 1.0)[0]
print("E(random init) %.3f  E(argmax) %.3f  E(truth) %.3f" % (E(init), E(np.argmax(lp, 1)), E(blk["labels_true"])))
for name, kw in (("random init, tol 0", dict(init=init, tol=0)), ("random init, tol 1000", dict(init=init, tol=1000)), ("argmax init, tol 0", dict(init=None, tol=0)),
                 ("argmax init, tol 1000", dict(init=None, tol=1000)), ("truth init, tol 0", dict(init=blk["labels_true"], tol=0)))[:int(os.environ.get("NCASE", "5"))]:
    if kw["init"] is not None:
        b.set_labels(kw["init"])
    t0 = time.time()
    res = b.solve(1.0, energy_tol_ppb=kw["tol"], init_mode=0 if kw["init"] is not None else 1)
    print("%-24s E %.3f rounds %d conv %s  %.2fs" % (name, E(b.get_labels()), res["rounds"], res["converged"], time.time() - t0), flush=True)
print("reference (seed 13, N 2000, K 10): swap via pygco 8705840.818, swap fine 8700903.203")
