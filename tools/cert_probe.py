"""Development experiment (GPU box; the arithmetic is NumPy): how many child strips of a coarse alpha-expansion could a ONE-HOP
FLOW certificate settle before the DP?  (HISTORY.md 3.1 item 6 (v).)  With every cell at "keep", a switch set S costs sum_S D + cut(S).
If every cell with D < 0 can ship its deficit -D to neighbouring cells with D > 0 along the edges (at most lambda per edge, at
most D_B into cell B in all), no switch set costs less than 0.  Split rule tested: a cell with surplus offers each of its
negative neighbours min(lambda, D_B / number of negative neighbours).  Prints, per scale, the share of 5 x 63 strips that
hold a negative cell (what the look lets through today) and that hold an UNCERTIFIED negative cell.
usage: python tools/cert_probe.py [workload] [block] [seed]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from phylo_hmrf_amd import Block, synthetic, workloads
from phylo_hmrf_amd.tree import PhyloTree
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
bi = int(sys.argv[2]) if len(sys.argv) > 2 else 7
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
blocks_def, S, K, nn, desc = workloads.workload(wl)
dev = torch.device("cuda", 0)
tree = PhyloTree(synthetic.tree_for(S)); rng = np.random.default_rng(seed)
P = synthetic.sample_ou_params(rng, tree, K); mu, cv = tree.mean_cov(P); cv = cv + 1e-3 * np.eye(S)
H, W, diag = blocks_def[bi]
n = workloads.block_nodes(H, W, diag); b = Block(n, S, K)
Xd = synthetic.device_observations(torch, dev, seed * 1000 + bi, H, W, diag, K, mu, cv); torch.cuda.synchronize()
b.set_observations_dev(Xd.data_ptr()); b.sync(); del Xd; b.build_grid_graph(H, W, diag, nn, 0.5)
cur = np.clip(P * (1 + 0.15 * rng.standard_normal(P.shape)), 1e-3, 50)
means, covars = tree.mean_cov(cur); covars = covars + 1e-3 * np.eye(S)
b.emission(means, covars)
r = b.solve(1.0, energy_tol_ppb=1000, init_mode=1, use_coarse=False)       # the fine moves to their tolerance: where the coarse rounds start
print("block %d: %d nodes, fine solve %d rounds, energy %.3f" % (bi, n, r["rounds"], r["energy"]))
DIRS = [(0, 1), (1, -1), (1, 0), (1, 1)]             # forward: E, SW, S, SE (lam[:, 0..3])
for s in (2, 4, 8):
    off = 0
    Hc, Wc = (H - 1 + off) // s + 1, (W - 1 + off) // s + 1
    tot = dict(strips=0, neg=0, unc=0, unc_opt=0, negcells=0, cells=0, certcells=0)
    for alpha in range(K):
        D, lam = b.coarse_problem(1.0, s, off, alpha)
        Dm = np.full((Hc + 2, Wc + 2), np.inf, dtype=np.float64)      # padded by one absent cell all round
        L = np.zeros((4, Hc + 2, Wc + 2))
        if diag:
            pos = 0
            for I in range(Hc):
                cnt = Wc - I
                Dm[I + 1, I + 1:Wc + 1] = D[pos:pos + cnt]
                for e in range(4):
                    L[e, I + 1, I + 1:Wc + 1] = lam[pos:pos + cnt, e]
                pos += cnt
            assert pos == D.size
        else:
            Dm[1:Hc + 1, 1:Wc + 1] = D.reshape(Hc, Wc)
            for e in range(4):
                L[e, 1:Hc + 1, 1:Wc + 1] = lam[:, e].reshape(Hc, Wc)
        present = np.isfinite(Dm)
        pinned = present & (Dm >= 1e29)
        neg = present & (Dm < 0)
        cap = np.where(pinned, np.inf, np.where(present & (Dm > 0), Dm, 0.0))
        # weight towards each of the 8 neighbours, seen from the cell
        def shift(A, di, dj):                        # A at (I + di, J + dj) brought to (I, J)
            out = np.zeros_like(A)
            src = A[max(di, 0):A.shape[0] + min(di, 0), max(dj, 0):A.shape[1] + min(dj, 0)]
            out[max(-di, 0):A.shape[0] + min(-di, 0), max(-dj, 0):A.shape[1] + min(-dj, 0)] = src
            return out
        Wd, nbneg = [], np.zeros(Dm.shape)
        for e, (di, dj) in enumerate(DIRS):
            Wd.append(((di, dj), L[e]))                                   # my forward edge
            Wd.append(((-di, -dj), shift(L[e], -di, -dj)))                # the backward neighbour's forward edge to me
        for (di, dj), w in Wd:
            nbneg += shift(neg.astype(float), di, dj) * (w > 0)
        recv = np.zeros(Dm.shape); recv_opt = np.zeros(Dm.shape)
        for (di, dj), w in Wd:                      # what the neighbour at (di, dj) offers me (I am one of its negative neighbours)
            capn = shift(cap, di, dj); nn_ = shift(nbneg, di, dj)
            share = np.where(np.isinf(capn), np.inf, capn / np.maximum(nn_, 1.0))
            recv += np.minimum(w, share)
            recv_opt += np.minimum(w, capn)
        cert = neg & (recv >= -Dm * 1.0001 + 1e-6)
        cert_opt = neg & (recv_opt >= -Dm)
        unc, unc_opt = neg & ~cert, neg & ~cert_opt
        core = (slice(1, Hc + 1), slice(1, Wc + 1))
        ng, uc, uo, pr = neg[core], unc[core], unc_opt[core], present[core]
        for r0 in range(0, Hc, 6):
            for c0 in range(0, Wc, 64):
                pw = pr[r0:r0 + 5, c0:c0 + 63]
                if not pw.any():
                    continue
                tot["strips"] += 1
                tot["neg"] += bool(ng[r0:r0 + 5, c0:c0 + 63].any())
                tot["unc"] += bool(uc[r0:r0 + 5, c0:c0 + 63].any())
                tot["unc_opt"] += bool(uo[r0:r0 + 5, c0:c0 + 63].any())
        tot["negcells"] += int(ng.sum()); tot["cells"] += int(pr.sum()); tot["certcells"] += int((ng & ~uc).sum())
    print("scale %d: %d strips x labels; with a negative cell %.1f%%; with an uncertified negative cell %.1f%% (no split of the capacities: %.1f%%); "
          "negative cells %.2f%% of cells, %.1f%% of them certified" % (s, tot["strips"], 100.0 * tot["neg"] / tot["strips"], 100.0 * tot["unc"] / tot["strips"],
           100.0 * tot["unc_opt"] / tot["strips"], 100.0 * tot["negcells"] / tot["cells"], 100.0 * tot["certcells"] / max(tot["negcells"], 1)))
