"""TEST INFRASTRUCTURE ONLY -- float64 NumPy model of the GPU label solver's moves.

The product's MRF E-step (phylo_hmrf_amd/csrc) replaces gco's alpha-beta swap with
energy-non-increasing moves that parallelise over nodes / chains:

  * ICM         : colour-ordered single-site minimisation,
  * chain moves : EXACT minimisation (Viterbi with the Potts min-trick) over every label of a whole
                  1-D chain of nodes (grid row / column / diagonal / anti-diagonal), all other
                  nodes fixed; chains whose nodes share no edge are solved simultaneously.

This module is the move-by-move model the HIP kernels are tested against (same tie-breaking:
lowest label wins, a chain keeps its incoming label on ties) and the CPU prototype used to
choose the move schedule against the gco oracle.  Energies follow SURVEY.md 8(a-E).
"""
import numpy as np


class Graph(object):
    """Symmetric CSR of an undirected weighted graph given as edges[E,2] (id1<id2), w[E]."""

    def __init__(self, n, edge_ids, w):
        a = np.asarray(edge_ids[:, 0], dtype=np.int64)
        b = np.asarray(edge_ids[:, 1], dtype=np.int64)
        w = np.asarray(w, dtype=np.float64)
        src = np.concatenate([a, b])
        dst = np.concatenate([b, a])
        ww = np.concatenate([w, w])
        order = np.lexsort((dst, src))
        self.n = int(n)
        self.src, self.col, self.wgt = src[order], dst[order], ww[order]
        self.row_ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(src, minlength=n), out=self.row_ptr[1:])
        self.edge_ids, self.w = np.stack([a, b], 1), w
        self.wtot = np.bincount(src, weights=ww, minlength=n)


def energy(g, unary, labels, beta):
    l = np.asarray(labels, dtype=np.int64)
    eu = float(np.sum(unary[np.arange(g.n), l]))
    ep = float(beta * np.sum(g.w * (l[g.edge_ids[:, 0]] != l[g.edge_ids[:, 1]])))
    return eu + ep, eu, ep


def neighbour_hist(g, labels, K, skip_a=None, skip_b=None, nodes=None):
    """hist[i,k] = sum_{j in N(i), l_j == k} w_ij, optionally skipping neighbours skip_a[i], skip_b[i]."""
    keep = np.ones(g.src.shape[0], dtype=bool)
    if skip_a is not None:
        keep &= g.col != skip_a[g.src]
    if skip_b is not None:
        keep &= g.col != skip_b[g.src]
    hist = np.zeros((g.n, K))
    np.add.at(hist, (g.src[keep], labels[g.col[keep]]), g.wgt[keep])
    return hist


# ------------------------------------------------------------------------------------------------
# grid geometry -> colours and chain families
# ------------------------------------------------------------------------------------------------
def grid_coords(H, W, diagonal):
    if diagonal:
        ii, jj = np.triu_indices(H)
    else:
        ii, jj = np.divmod(np.arange(H * W), W)
    return ii.astype(np.int64), jj.astype(np.int64)


def icm_colours(H, W, diagonal):
    ii, jj = grid_coords(H, W, diagonal)
    return (ii % 2) * 2 + (jj % 2), 4


def chain_families(H, W, diagonal, num_neighbor=8):
    """List of families; a family = (chains, colour_of_chain, n_colours) where chains is a list of
    node-id arrays in chain order.  Chains of one colour share no edge with each other."""
    ii, jj = grid_coords(H, W, diagonal)
    n = ii.shape[0]
    ids = np.arange(n)
    fams = []

    def build(key_chain, key_pos, colour_of_key, ncol):
        order = np.lexsort((key_pos, key_chain))
        kc = key_chain[order]
        cuts = np.flatnonzero(np.diff(kc)) + 1
        chains = np.split(ids[order], cuts)
        keys = kc[np.concatenate([[0], cuts])]
        return chains, colour_of_key(keys), ncol

    fams.append(build(ii, jj, lambda k: k % 2, 2))                       # rows
    fams.append(build(jj, ii, lambda k: k % 2, 2))                       # columns
    if num_neighbor == 8:
        fams.append(build(jj - ii, ii, lambda k: k % 3, 3))              # diagonals  (x+1, y+1)
        fams.append(build(ii + jj, ii, lambda k: k % 3, 3))              # anti-diagonals (x+1, y-1)
    return fams


def pack_family(fam):
    """-> nodes[C, L] (-1 padded), lens[C], colour[C]."""
    chains, colour, ncol = fam
    C = len(chains)
    L = max(len(c) for c in chains)
    nodes = -np.ones((C, L), dtype=np.int64)
    lens = np.zeros(C, dtype=np.int64)
    for c, ch in enumerate(chains):
        nodes[c, :len(ch)] = ch
        lens[c] = len(ch)
    return nodes, lens, np.asarray(colour), ncol


# ------------------------------------------------------------------------------------------------
# moves
# ------------------------------------------------------------------------------------------------
def icm_sweep(g, unary, labels, beta, colours, ncol):
    """One colour-ordered ICM sweep, in place.  cost_k = unary_k - beta*hist_k ; argmin, lowest k on ties,
    but the current label is kept unless strictly improved."""
    K = unary.shape[1]
    changed = 0
    for c in range(ncol):
        idx = np.flatnonzero(colours == c)
        hist = neighbour_hist(g, labels, K)[idx]
        cost = unary[idx] - beta * hist
        best = np.argmin(cost, axis=1)
        cur = labels[idx]
        better = cost[np.arange(len(idx)), best] < cost[np.arange(len(idx)), cur]
        changed += int(np.sum(better))
        labels[idx] = np.where(better, best, cur)
    return changed


def link_weights(g, nodes):
    """w(nodes[c,t], nodes[c,t+1]) or 0 when there is no such edge / padding."""
    C, L = nodes.shape
    a, b = nodes[:, :-1], nodes[:, 1:]
    key = {}
    lw = np.zeros((C, max(L - 1, 0)))
    # vectorised lookup via sorted (src*n + col) keys
    keys = g.src * g.n + g.col
    q = np.where((a >= 0) & (b >= 0), a * g.n + b, -1)
    pos = np.searchsorted(keys, q.ravel())
    pos = np.clip(pos, 0, len(keys) - 1)
    hit = keys[pos] == q.ravel()
    lw = np.where(hit, g.wgt[pos], 0.0).reshape(q.shape)
    return lw


def chain_move(g, unary, labels, beta, nodes, lens, sel):
    """Exact re-labelling of the selected chains (rows of `nodes` where sel), in place.
    Returns the number of nodes whose label changed."""
    K = unary.shape[1]
    nd = nodes[sel]
    ln = lens[sel]
    C, L = nd.shape
    if C == 0:
        return 0
    valid = nd >= 0
    safe = np.where(valid, nd, 0)
    prev = -np.ones(g.n, dtype=np.int64)
    nxt = -np.ones(g.n, dtype=np.int64)
    prev[nd[:, 1:][valid[:, 1:]]] = nd[:, :-1][valid[:, 1:]]
    nxt[nd[:, :-1][valid[:, 1:]]] = nd[:, 1:][valid[:, 1:]]
    hist = neighbour_hist(g, labels, K, prev, nxt)
    theta = unary[safe] - beta * hist[safe]                # [C, L, K]  (constant beta*Wtot dropped)
    theta = np.where(valid[:, :, None], theta, 0.0)
    lw = beta * link_weights(g, nd)                        # [C, L-1]
    # forward: m_t[k] = theta_t[k] + min(m_{t-1}[k], min_j m_{t-1}[j] + c_{t-1})
    m = theta[:, 0, :].copy()
    jump = np.zeros((C, L, K), dtype=bool)                 # True: best predecessor is the global argmin
    amin = np.zeros((C, L), dtype=np.int64)
    for t in range(1, L):
        mm = m.min(axis=1)
        am = m.argmin(axis=1)
        alt = mm + lw[:, t - 1]
        jp = alt[:, None] < m                              # strict: ties keep the same label
        m_new = theta[:, t, :] + np.where(jp, alt[:, None], m)
        live = valid[:, t]
        m = np.where(live[:, None], m_new, m)
        jump[:, t, :] = jp & live[:, None]
        amin[:, t] = am
    # backtrack
    new = np.zeros((C, L), dtype=np.int64)
    cur = m.argmin(axis=1)
    for t in range(L - 1, -1, -1):
        live = valid[:, t]
        new[:, t] = cur
        if t > 0:
            jp = jump[np.arange(C), t, cur]
            cur = np.where(live & jp, amin[:, t], cur)
    old = labels[safe]
    ch = int(np.sum((new != old) & valid))
    labels[nd[valid]] = new[valid]
    return ch


def family_sweep(g, unary, labels, beta, packed):
    nodes, lens, colour, ncol = packed
    ch = 0
    for c in range(ncol):
        ch += chain_move(g, unary, labels, beta, nodes, lens, colour == c)
    return ch


def solve(g, unary, init, beta, H=None, W=None, diagonal=None, num_neighbor=8, max_rounds=50, use_chains=True,
          trace=None):
    """ICM + chain moves until a full round changes nothing (or max_rounds)."""
    labels = np.asarray(init, dtype=np.int64).copy()
    geom = H is not None
    if geom:
        colours, ncol = icm_colours(H, W, diagonal)
        fams = [pack_family(f) for f in chain_families(H, W, diagonal, num_neighbor)] if use_chains else []
    else:
        raise NotImplementedError("prototype needs grid geometry")
    for r in range(max_rounds):
        ch = 0
        for p in fams:
            ch += family_sweep(g, unary, labels, beta, p)
        ch += icm_sweep(g, unary, labels, beta, colours, ncol)
        if trace is not None:
            trace.append((r, ch, energy(g, unary, labels, beta)[0]))
        if ch == 0:
            break
    return labels
