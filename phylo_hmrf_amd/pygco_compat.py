"""Drop-in for the one pygco call on the hot path (phylo_hmrf.py:496-498):

    labels = pygco.cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, n_iter=5000,
                                     algorithm='swap', init_labels=..., down_weight_factor=None)

Same argument names, order and return (int32 labels [n]); the labelling is computed by the GPU label solver on the
float energy (no integer quantisation, so `down_weight_factor` has no effect).  Only Potts `pairwise_cost`
(= beta * (1 - delta), which is what phylo_hmrf.py:524-536 builds) is accepted; anything else raises ValueError.
A maintainer switches the reference over with `import phylo_hmrf_amd.pygco_compat as pygco`.

The reference passes no geometry, only the edge list -- but its edge lists are the sorted stencil edges of a contact-map
block (utility.py:1871-2053: a full H x W block row-major, or the upper triangle of an N x N block row-major, 8- or
4-neighbour), and the solver's chain / strip moves need that geometry.  `infer_grid` recovers (H, W, diagonal,
num_neighbor) from the edge list; the library then checks every edge against it (phmrf_block_set_grid).  A graph that is
not such a grid is solved with the general-graph moves only (ICM, component and path moves, alpha-expansion by a minimum
cut on the device: maxflow.hip) and a RuntimeWarning says so
(`strict_grid=True` raises instead).
"""
import warnings

import numpy as np

from ._lib import PhmrfError
from .block import Block

ERR_INVALID = 1


def grid_candidates(n, edges):
    """Geometries (H, W, diagonal, num_neighbor) compatible with the node count and the neighbours of node 0 -- the
    top-left cell of either layout: its neighbours are {1, W[, W+1]} in a full block and {1[, N]} in a diagonal block."""
    edges = np.asarray(edges)
    if edges.shape[0] == 0:
        return []
    a, b = np.int64(edges[:, 0]), np.int64(edges[:, 1])
    nb0 = np.unique(np.concatenate([b[a == 0], a[b == 0]]))
    cands = []
    N = int(round((np.sqrt(8.0 * n + 1.0) - 1.0) / 2.0))
    if N * (N + 1) // 2 == n and N >= 1:
        if nb0.tolist() in ([1], []) or n == 1:
            cands.append((N, N, True, 4))
        if nb0.tolist() in ([1, N], [1], [N]) and N > 1:
            cands.append((N, N, True, 8))
    widths = set()
    if nb0.tolist() == [1]:
        widths.add(n)                                     # a single row (a single column is the same chain)
    for v in nb0.tolist():
        if v > 1:
            widths.update((v, v - 1))
    for W in sorted(widths):
        if W >= 1 and n % W == 0:
            H = n // W
            cands.append((H, W, False, 4))
            cands.append((H, W, False, 8))
    return cands


def infer_grid(block, n, edges):
    """Declare the first candidate geometry that the library accepts for the block's edge list -> the geometry or None."""
    for cand in grid_candidates(n, edges):
        try:
            block.set_grid(*cand)
            return cand
        except PhmrfError as e:
            if e.status != ERR_INVALID:
                raise
    return None


def cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, n_iter=-1, algorithm="expansion",
                      init_labels=None, down_weight_factor=None, grid=None, strict_grid=False):
    """grid: optional (H, W, diagonal, num_neighbor) of the block; by default it is inferred from the edge list."""
    unary_cost = np.asarray(unary_cost, dtype=np.float64)
    pairwise_cost = np.asarray(pairwise_cost, dtype=np.float64)
    n, K = unary_cost.shape
    if pairwise_cost.shape != (K, K):
        raise ValueError("pairwise_cost must be [n_labels, n_labels]")
    beta = pairwise_cost[0, 1] if K > 1 else 0.0
    potts = np.full((K, K), beta)
    np.fill_diagonal(potts, 0.0)
    if not np.allclose(pairwise_cost, potts):
        raise ValueError("only the Potts pairwise cost beta*(1-delta) of the reference is supported")
    if algorithm not in ("swap", "expansion"):
        raise ValueError("algorithm must be 'swap' or 'expansion'")
    edges = np.asarray(edges)
    b = Block(n, 1, K)
    try:
        b.set_graph(np.int64(edges[:, 0:2]), np.asarray(edge_weights, dtype=np.float64))
        if grid is not None:
            b.set_grid(*grid)
        elif infer_grid(b, n, edges) is None:
            msg = ("cut_general_graph: the edge list is not the stencil of a contact-map block (utility.py:1871-2053); "
                   "solving with general-graph moves only (ICM, component and path moves, alpha-expansion by minimum cut): "
                   "host-synchronous and slower per node than the grid path; energy parity with gco's swap off the grid is "
                   "tested on k-NN graphs")
            if strict_grid:
                raise ValueError(msg)
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
        b.set_logprob(-unary_cost)
        if init_labels is not None:
            b.set_labels(np.asarray(init_labels))
            init_mode = 0
        else:
            init_mode = 1
        rounds = 64 if n_iter is None or n_iter < 0 else max(1, min(int(n_iter), 64))
        b.solve_fast(float(beta), max_rounds=rounds, init_mode=init_mode)
        return b.get_labels()
    finally:
        b.close()
