"""Drop-in for the one pygco call on the hot path (phylo_hmrf.py:496-498):

    labels = pygco.cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, n_iter=5000,
                                     algorithm='swap', init_labels=..., down_weight_factor=None)

Same argument names, order and return (int32 labels [n]); the labelling is computed by the GPU label solver on the
float energy (no integer quantisation, so `down_weight_factor` has no effect).  Only Potts `pairwise_cost`
(= beta * (1 - delta), which is what phylo_hmrf.py:524-536 builds) is accepted; anything else raises ValueError.
A maintainer switches the reference over with `import phylo_hmrf_amd.pygco_compat as pygco`.
"""
import numpy as np

from .block import Block


def cut_general_graph(edges, edge_weights, unary_cost, pairwise_cost, n_iter=-1, algorithm="expansion",
                      init_labels=None, down_weight_factor=None, grid=None):
    """grid: optional (H, W, diagonal, num_neighbor) of the block, enabling the chain and strip moves."""
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
