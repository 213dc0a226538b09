"""Development helper (GPU only): where does the solver's labelling differ from gco's (fine quantisation) on the cfg2 block?"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from scipy import ndimage
from oracle import gco_ref, ref_numpy as R, synth
from phylo_hmrf_amd import Block
seed, N, K = 13, 2000, 10
blk = synth.make_block(seed, N, N, 4, K, True)
X = blk["X"]; n = X.shape[0]
w, eid = R.edge_weights_from_distance(blk["edges"], 0.5)
lp = R.log_multivariate_normal_density_full(X, blk["means"], blk["covars"])
init = np.random.default_rng(seed + 7).integers(0, K, n)
E = lambda lab: R.mrf_energy(lab, lp, eid, w, 1.0)[0]
b = Block(n, 4, K); b.set_graph(eid, w); b.set_grid(N, N, True, 8); b.set_logprob(lp)
res = b.solve(1.0, energy_tol_ppb=0, init_mode=1)
L = b.get_labels()
print("argmax start: E %.3f" % E(L), flush=True)
# extra passes at other cut geometries from the fixed point
rng = np.random.default_rng(1)
ch_tot = 0
for rep in range(40):
    o = rep % 2; sr = int(rng.integers(0, 6)); sc = int(rng.integers(0, 64))
    for a in range(-1, K):
        ch_tot += b.strip_pass(1.0, o, sr, sc, a)
    if rep % 10 == 9:
        print("after %d extra random-cut sweeps: E %.3f (changed %d)" % (rep + 1, E(b.get_labels()), ch_tot), flush=True)
res = b.solve(1.0, energy_tol_ppb=0)
L2 = b.get_labels()
print("then solve again: E %.3f" % E(L2), flush=True)
t0 = time.time()
Lg = gco_ref.cut_general_graph(eid, w, -lp, R.potts_matrix(K, 1.0), n_iter=5000, algorithm="swap", init_labels=init, quant="fine")
print("gco fine: E %.3f (%.0fs)" % (E(Lg), time.time() - t0), flush=True)
ii, jj = np.triu_indices(N)
def img(v, fill=-1):
    m = np.full((N, N), fill, dtype=np.int32); m[ii, jj] = v; return m
for name, La in (("argmax-start", L), ("after extra sweeps", L2)):
    diff = img((La != Lg).astype(np.int32), 0)
    lab_img, ncomp = ndimage.label(diff, structure=np.ones((3, 3)))
    sl = ndimage.find_objects(lab_img)
    area = ndimage.sum(diff, lab_img, index=np.arange(1, ncomp + 1))
    hh = np.array([s[0].stop - s[0].start for s in sl]); ww = np.array([s[1].stop - s[1].start for s in sl])
    print("%s vs gco fine: %d differing nodes in %d connected regions" % (name, int(diff.sum()), ncomp))
    md = np.minimum(hh, ww)
    for lo, hi in ((1, 2), (2, 4), (4, 6), (6, 10), (10, 20), (20, 10**9)):
        k = (md >= lo) & (md < hi)
        print("   min(bbox h, w) in [%d, %d): %d regions, %d nodes" % (lo, hi, int(k.sum()), int(area[k].sum())))
    # energy if we took gco's labels inside the big regions only
    big = np.isin(lab_img, np.flatnonzero(md >= 6) + 1)[ii, jj]
    Lmix = np.where(big, Lg, La)
    print("   E(ours) %.3f  E(ours with gco's labels in regions of min-dim >= 6) %.3f  E(gco) %.3f" % (E(La), E(Lmix), E(Lg)))
