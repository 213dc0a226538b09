"""Host edge-list builder for grid blocks in the reference's on-disk form [E,3] = (id1, id2, d), sorted by
(id1, id2) -- what `edgelist.<res>Kb.observed.<run>.npy` holds (utility.py:1871-2053, phylo_hmrf.py:1700-1701).
Used only to WRITE caches (synthetic data); the fit itself can build the graph on the device
(phmrf_block_build_grid_graph)."""
import numpy as np


def grid_edges(X, H, W, diagonal, num_neighbor=8):
    X = np.asarray(X, dtype=np.float64)
    stencil = [(0, 1), (1, 1), (1, 0), (1, -1)] if num_neighbor == 8 else [(0, 1), (1, 0)]   # utility.py:1898-1905
    if diagonal:
        ii, jj = np.triu_indices(H)
        lut = np.full((H, W), -1, dtype=np.int64)
        lut[ii, jj] = np.arange(ii.shape[0])
    else:
        ii, jj = np.divmod(np.arange(H * W), W)
        lut = np.arange(H * W, dtype=np.int64).reshape(H, W)
    nrm = np.sqrt((X * X).sum(axis=1))
    rows = []
    for di, dj in stencil:
        i2, j2 = ii + di, jj + dj
        ok = (i2 >= 0) & (i2 < H) & (j2 >= 0) & (j2 < W)
        if diagonal:
            ok &= i2 <= j2                                                     # utility.py:1917
        a = np.flatnonzero(ok)
        b = lut[i2[a], j2[a]]
        d = ((X[a] - X[b]) ** 2).sum(axis=1) / (nrm[a] * nrm[b] + 1e-16)       # utility.py:1935-1939
        if diagonal:
            d = np.where((ii[a] == jj[a]) & (i2[a] == j2[a]), 0.5 * d, d)      # utility.py:1942-1953
        rows.append(np.column_stack([a, b, d]))
    e = np.concatenate(rows, axis=0)
    return e[np.lexsort((e[:, 1], e[:, 0]))]
