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


# ------------------------------------------------------------------------------------------------
# connected-component relabel moves (model of csrc/moves.hip: component pass)
# ------------------------------------------------------------------------------------------------
def component_pass(g, unary, labels, beta, margin=True):
    """Relabel whole connected components of equal label: for every component C and label k the exact
    dE = sum_{i in C}(u_i(k) - u_i(cur)) - beta * sum_{boundary edges to label k} w; a component moves to its
    best strictly-improving label unless an adjacent candidate component has a better gain (ties: lower
    root id).  In place; returns the number of relabelled nodes."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import connected_components
    K = unary.shape[1]
    a, b = g.edge_ids[:, 0], g.edge_ids[:, 1]
    same = labels[a] == labels[b]
    A = sp.coo_matrix((np.ones(int(same.sum())), (a[same], b[same])), shape=(g.n, g.n))
    nc, comp0 = connected_components(A, directed=False)
    # root id = smallest node id of the component (what the device's min-propagation converges to)
    root_of = np.full(nc, g.n, dtype=np.int64)
    np.minimum.at(root_of, comp0, np.arange(g.n))
    U = np.zeros((nc, K))
    np.add.at(U, comp0, unary)
    cs, cd = comp0[g.src], comp0[g.col]
    bd = cs != cd
    np.add.at(U, (cs[bd], labels[g.col[bd]]), -beta * g.wgt[bd])
    cur = np.zeros(nc, dtype=np.int64)
    cur[comp0] = labels
    tc = U[np.arange(nc), cur]
    best = np.argmin(U, axis=1)
    bv = U[np.arange(nc), best]
    mar = (1e-5 * np.abs(tc) + 1e-6) if margin else 0.0
    cand = bv < tc - mar
    gain = np.where(cand, bv - tc, 0.0)
    blocked = np.zeros(nc, dtype=bool)
    ci, cj = cs[bd], cd[bd]
    worse = cand[ci] & ((gain[cj] < gain[ci]) | ((gain[cj] == gain[ci]) & (root_of[cj] < root_of[ci]) & cand[cj]))
    blocked[ci[worse]] = True
    move = cand & ~blocked
    mv = move[comp0]
    labels[mv] = best[comp0[mv]]
    return int(mv.sum())


# ------------------------------------------------------------------------------------------------
# strip fusion moves (model of csrc/strip.hip)
# ------------------------------------------------------------------------------------------------
STRIP_H = 5          # rows per strip: profile of H+1 = 6 binary cells = 64 states = one wavefront
STRIP_L = 63         # columns per strip segment


def best_alternative(g, unary, labels, beta):
    """proposal p_i = argmin_{k != l_i} (u_i(k) - beta * sum_{j in N(i), l_j == k} w_ij), lowest k on ties."""
    K = unary.shape[1]
    cost = unary - beta * neighbour_hist(g, labels, K)
    cost[np.arange(g.n), labels] = np.inf
    return np.argmin(cost, axis=1)


def strip_node_table(H, W, diagonal, orient, shift_r, shift_c):
    """-> nodes[NS, STRIP_L*STRIP_H] (column-major cells, -1 missing), ncols[NS].  Geometry (shared with the
    kernel): orient 0: strip rows = grid rows; orient 1: strip rows = grid columns.  Bands of STRIP_H strip-rows
    start at b*(STRIP_H+1) - shift_r and are separated by one fixed strip-row; segments of STRIP_L strip-columns
    start at s*64 - shift_c and are separated by one fixed strip-column."""
    h, L = STRIP_H, STRIP_L
    if diagonal:
        idmap = -np.ones((H, W), dtype=np.int64)
        ii, jj = np.triu_indices(H)
        idmap[ii, jj] = np.arange(ii.shape[0])
    else:
        idmap = np.arange(H * W).reshape(H, W)
    mp = idmap if orient == 0 else idmap.T
    Hs, Ws = mp.shape
    nb = -(-(Hs + shift_r) // (h + 1))
    ns = -(-(Ws + shift_c) // 64)
    out, ncols = [], []
    for b in range(nb):
        rs0 = b * (h + 1) - shift_r
        for s in range(ns):
            cs0 = s * 64 - shift_c
            ca, cb = max(cs0, 0), min(cs0 + L, Ws)
            if cb <= ca:
                continue
            blk = -np.ones((L, h), dtype=np.int64)
            for rr in range(h):
                r = rs0 + rr
                if 0 <= r < Hs:
                    blk[:cb - ca, rr] = mp[r, ca:cb]
            out.append(blk.reshape(-1))
            ncols.append(cb - ca)
    return np.array(out), np.array(ncols)


def strip_fusion(g, unary, labels, prop, beta, H, W, diagonal, orient, shift_r, shift_c):
    """Exact binary fusion x_i in {keep l_i, take prop_i} over every strip simultaneously (profile DP over
    2^(STRIP_H+1) states).  In place; returns the number of changed nodes."""
    h = STRIP_H
    nodes, ncols = strip_node_table(H, W, diagonal, orient, shift_r, shift_c)
    NS, T = nodes.shape
    n = g.n
    valid = nodes >= 0
    safe = np.where(valid, nodes, 0)
    sid = -np.ones(n, dtype=np.int64)
    ss, pp = np.nonzero(valid)
    sid[nodes[ss, pp]] = ss
    lab = np.asarray(labels, dtype=np.int64)
    prop = np.asarray(prop, dtype=np.int64)
    inside = (sid[g.src] >= 0) & (sid[g.src] == sid[g.col])
    c0 = unary[np.arange(n), lab].copy()
    c1 = unary[np.arange(n), prop].copy()
    o = ~inside
    np.add.at(c0, g.src[o], beta * g.wgt[o] * (lab[g.src[o]] != lab[g.col[o]]))
    np.add.at(c1, g.src[o], beta * g.wgt[o] * (prop[g.src[o]] != lab[g.col[o]]))
    keys = g.src * n + g.col

    def wlook(a, b):
        q = np.where((a >= 0) & (b >= 0), a * n + b, -1)
        p = np.clip(np.searchsorted(keys, q.ravel()), 0, len(keys) - 1)
        hit = keys[p] == q.ravel()
        return np.where(hit, g.wgt[p], 0.0).reshape(q.shape)

    rr = np.arange(T) % h

    def shifted(off, ok):
        res = -np.ones_like(nodes)
        if off < T:
            res[:, off:] = nodes[:, :T - off]
        res[:, ~ok] = -1
        return res

    nbs = [(shifted(1, rr > 0), 0), (shifted(h + 1, rr > 0), h), (shifted(h, rr >= 0), h - 1),
           (shifted(h - 1, rr < h - 1), h - 2)]
    ws = [wlook(nodes, nb) for nb, _ in nbs]
    labv = np.where(valid, lab[safe], 0)
    prv = np.where(valid, prop[safe], 0)
    BIG = 1e30
    C0 = np.where(valid, c0[safe], 0.0)
    C1 = np.where(valid & (prv != labv), c1[safe], BIG)
    NSt = 1 << (h + 1)
    st = np.arange(NSt)
    bbit = st & 1
    m = np.zeros((NS, NSt))
    back = np.zeros((NS, T, NSt), dtype=bool)
    ncell = ncols * h
    s_fin = np.zeros(NS, dtype=np.int64)
    for t in range(T):
        def total(pred):
            cost = np.where(bbit[None, :] == 1, C1[:, t][:, None], C0[:, t][:, None])
            labi = np.where(bbit[None, :] == 1, prv[:, t][:, None], labv[:, t][:, None])
            for (nb, posn), w in zip(nbs, ws):
                nbv = nb[:, t] >= 0
                nsafe = np.where(nbv, nb[:, t], 0)
                bitj = (pred >> posn) & 1
                labj = np.where(bitj[None, :] == 1, prop[nsafe][:, None], lab[nsafe][:, None])
                cost = cost + np.where(nbv[:, None], beta * w[:, t][:, None] * (labi != labj), 0.0)
            return m[:, pred] + cost
        a0 = total(st >> 1)
        a1 = total((st >> 1) | (1 << h))
        # ties keep the predecessor whose leaving bit equals the new cell's bit (the kernel's "self" lane)
        take1 = np.where(bbit[None, :] == 1, a1 <= a0, a1 < a0)
        m = np.where(take1, a1, a0)
        back[:, t, :] = take1
        done = ncell == t + 1
        s_fin[done] = np.argmin(m[done], axis=1)       # the kernel stops at the strip's own last cell
    s = s_fin.copy()
    x = np.zeros((NS, T), dtype=np.int64)
    for t in range(T - 1, -1, -1):
        live = t < ncell
        x[:, t] = np.where(live, s & 1, 0)
        d = back[np.arange(NS), t, s]
        s = np.where(live, (s >> 1) | (d.astype(np.int64) << h), s)
    newlab = np.where(x == 1, prv, labv)
    ch = int(((newlab != labv) & valid).sum())
    labels[nodes[valid]] = newlab[valid]
    return ch


def peel(g, unary, labels, beta, H, W, diagonal, orient, shift_r, shift_c, alpha, max_sweeps=None):
    """The exact filter in front of the strip alpha-expansions (model of strip_cols_kernel's sweeps).

    s_i = cost of switching node i alone to alpha;  disc_ij = beta w_ij (2 - [l_i != l_j]) = what an edge inside a
    switching set saves against the single-site sums.  For an optimal switching set C* of a strip:
      every member:   s_i <= sum_{j in C*, j ~ i} disc_ij          (else i would leave at a profit)
      if it improves: s_i <  1/2 sum_{j in C*, j ~ i} disc_ij  for some member (a seed)
    Deleting, sweep after sweep, the nodes of U (initially every strip node with l_i != alpha) that violate the first
    line with U in place of C* never deletes a member of C*.  Returns (U[n] bool, seeded[NS] bool): the expansion of
    alpha can only move nodes of U, and only in strips with a seed.  max_sweeps None: to the fixed point."""
    nodes, _ = strip_node_table(H, W, diagonal, orient, shift_r, shift_c)
    n = g.n
    sid = -np.ones(n, dtype=np.int64)
    ss, pp = np.nonzero(nodes >= 0)
    sid[nodes[ss, pp]] = ss
    NS = nodes.shape[0]
    lab = np.asarray(labels, dtype=np.int64)
    li, lj = lab[g.src], lab[g.col]
    s = unary[:, alpha] - unary[np.arange(n), lab]
    np.add.at(s, g.src, beta * g.wgt * ((lj != alpha).astype(np.float64) - (lj != li)))
    disc = beta * g.wgt * (2.0 - (lj != li))
    same = (sid[g.src] >= 0) & (sid[g.src] == sid[g.col])
    U = (sid >= 0) & (lab != alpha) & (unary[:, alpha] < 1e29)
    sweeps = 0
    while True:
        e = same & U[g.col]
        cap = np.zeros(n)
        np.add.at(cap, g.src[e], disc[e])
        seeds = U & (s <= cap) & (s < 0.5 * cap)
        U2 = U & (s <= cap)
        sweeps += 1
        done = bool((U2 == U).all()) or (max_sweeps is not None and sweeps >= max_sweeps)
        U = U2
        if done:
            break
    seeded = np.zeros(NS, dtype=bool)
    np.logical_or.at(seeded, sid[seeds], True)
    return U, seeded


# ------------------------------------------------------------------------------------------------
# coarse alpha-expansion (model of phylo_hmrf_amd/csrc/coarse.hip)
# ------------------------------------------------------------------------------------------------
def coarse_problem(g, unary, labels, beta, H, W, diagonal, s, off, alpha):
    """Super-cells of s x s nodes (super-cell of node (i, j) = ((i + off) // s, (j + off) // s)) keep their labels or
    switch to alpha as a whole:  dE(x) = sum_A D_A x_A + beta * sum_AB lam_AB [x_A != x_B].
    -> (D[nc] with beta folded in, lam[nc,4] forward pair weights E, SW, S, SE (without beta), cnode[n] super-cell of each
        node, Hc, Wc).  Coarse node ids follow the fine layout (row-major; upper triangle row-major for a diagonal block)."""
    ii, jj = grid_coords(H, W, diagonal)
    I, J = (ii + off) // s, (jj + off) // s
    Hc, Wc = (H - 1 + off) // s + 1, (W - 1 + off) // s + 1
    cnode = (I * Wc - (I * (I - 1)) // 2 + (J - I)) if diagonal else I * Wc + J
    nc = Hc * (Hc + 1) // 2 if diagonal else Hc * Wc
    lab = np.asarray(labels, dtype=np.int64)
    n = g.n
    D = np.bincount(cnode, weights=unary[np.arange(n), alpha] - unary[np.arange(n), lab], minlength=nc)
    a, b = g.edge_ids[:, 0], g.edge_ids[:, 1]
    la, lb = lab[a], lab[b]
    t00, t01, t10 = g.w * (la != lb), g.w * (la != alpha), g.w * (alpha != lb)
    ca, cb = cnode[a], cnode[b]
    same = ca == cb
    D -= beta * np.bincount(ca[same], weights=t00[same], minlength=nc)
    x = ~same
    lv = 0.5 * (t10[x] + t01[x] - t00[x])
    D += beta * (np.bincount(ca[x], weights=t10[x] - t00[x] - lv, minlength=nc) +
                 np.bincount(cb[x], weights=t01[x] - t00[x] - lv, minlength=nc))
    lam = np.zeros((nc, 4))
    dI, dJ = I[b[x]] - I[a[x]], J[b[x]] - J[a[x]]
    # the pair's slot belongs to the upper / left super-cell: E (0,1) -> 0, SW (1,-1) -> 1, S (1,0) -> 2, SE (1,1) -> 3;
    # a fine edge whose far end lies in the previous coarse column of the same coarse row is the E pair of that one
    fwd = ~((dI == 0) & (dJ == -1))
    slot = np.where(dI == 0, 0, 2 + dJ)
    np.add.at(lam, (ca[x][fwd], slot[fwd]), lv[fwd])
    np.add.at(lam, (cb[x][~fwd], 0), lv[~fwd])
    return D, lam, cnode, Hc, Wc


def coarse_cross_weight(g, H, W, diagonal, s, off):
    """Total weight of the fine edges that leave each super-cell: a super-cell with D > beta * this is in no optimal switch
    set (lambda <= w on every cross edge), the kernel pins it."""
    ii, jj = grid_coords(H, W, diagonal)
    I, J = (ii + off) // s, (jj + off) // s
    Wc = (W - 1 + off) // s + 1
    Hc = (H - 1 + off) // s + 1
    cnode = (I * Wc - (I * (I - 1)) // 2 + (J - I)) if diagonal else I * Wc + J
    nc = Hc * (Hc + 1) // 2 if diagonal else Hc * Wc
    a, b = g.edge_ids[:, 0], g.edge_ids[:, 1]
    x = cnode[a] != cnode[b]
    return np.bincount(cnode[a][x], weights=g.w[x], minlength=nc) + np.bincount(cnode[b][x], weights=g.w[x], minlength=nc)


def coarse_expansion(g, unary, labels, beta, H, W, diagonal, s, off, alpha, shift_r=0, shift_c=0):
    """One coarse alpha-expansion: the super-cell problem solved by one strip pass per orientation (strip_fusion on the
    coarse grid, all super-cells start at 'keep'), then applied.  In place; returns the number of changed nodes."""
    D, lam, cnode, Hc, Wc = coarse_problem(g, unary, labels, beta, H, W, diagonal, s, off, alpha)
    nc = D.shape[0]
    ci, cj = grid_coords(Hc, Wc, diagonal)
    if diagonal:
        idmap = -np.ones((Hc, Wc), dtype=np.int64)
        idmap[ci, cj] = np.arange(nc)
    else:
        idmap = np.arange(nc).reshape(Hc, Wc)
    e, w = [], []
    for q, (di, dj) in enumerate(((0, 1), (1, -1), (1, 0), (1, 1))):
        i2, j2 = ci + di, cj + dj
        ok = (i2 < Hc) & (j2 >= 0) & (j2 < Wc)
        ok[ok] &= idmap[i2[ok], j2[ok]] >= 0
        src = np.flatnonzero(ok)
        e.append(np.stack([src, idmap[i2[src], j2[src]]], 1))
        w.append(lam[src, q])
    e, w = np.concatenate(e), np.concatenate(w)
    keep = w != 0
    gc = Graph(nc, e[keep], w[keep])
    uc = np.stack([np.zeros(nc), D], 1)
    xl = np.zeros(nc, dtype=np.int64)
    strip_fusion(gc, uc, xl, np.ones(nc, dtype=np.int64), beta, Hc, Wc, diagonal, 0, shift_r % 6, shift_c % 64)
    strip_fusion(gc, uc, xl, np.ones(nc, dtype=np.int64), beta, Hc, Wc, diagonal, 1, (shift_r + 3) % 6, (shift_c + 31) % 64)
    lab = np.asarray(labels)
    sel = (xl[cnode] == 1) & (lab != alpha)
    labels[sel] = alpha
    return int(sel.sum())
