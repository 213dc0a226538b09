"""TEST / BENCH INFRASTRUCTURE: seeded synthetic multi-species Hi-C blocks.

Per block: a ground-truth label image made of random rectangles (mean run ~25 bins) over K states;
per state OU parameters sampled in the reference's feasible box (beta, lambda in [0.05, 2],
theta in [0, 4], root variance in [0.1, 1]; bounds phylo_hmrf.py:1365-1366, :1412-1413) and turned
into (mu_k, Sigma_k) by the a2 recursion (phylo_hmrf.py:985-1036); x_i ~ N(mu_{l_i}, Sigma_{l_i})
clipped at 0 (the reference's features are log(1+x) >= 0, utility.py:363); edges and distances
d_ij as utility.py:1871-2053.  Only numpy is used so the generator also runs on the GPU box.
"""
import numpy as np

from . import ref_numpy as R

TREE4 = [[0, 1], [1, 2], [1, 3], [3, 4], [4, 5], [4, 6], [3, 7]]           # example_input/edge.1.txt
TREE8 = [[0, 1], [0, 2], [1, 3], [1, 4], [2, 5], [2, 6], [3, 7], [3, 8], [4, 9], [4, 10],
         [5, 11], [5, 12], [6, 13], [6, 14]]                                # balanced ladder, 8 leaves


def tree_for(S):
    if S == 4:
        return TREE4
    if S == 8:
        return TREE8
    raise ValueError("synthetic trees are defined for S in {4, 8}")


def sample_ou_params(rng, tt, K):
    B = tt.branch_dim
    P = np.zeros((K, tt.n_params))
    P[:, 0] = rng.uniform(0.1, 1.0, K)
    P[:, 1:1 + B] = rng.uniform(0.05, 2.0, (K, B))
    P[:, 1 + B:1 + 2 * B] = rng.uniform(0.05, 2.0, (K, B))
    P[:, 1 + 2 * B:] = rng.uniform(0.0, 4.0, (K, B + 1))
    return P


def label_image(rng, H, W, K, mean_run=25):
    """Random rectangles painted over a background; later rectangles overwrite earlier ones."""
    img = rng.integers(0, K, size=(H, W)).astype(np.int32) if False else np.zeros((H, W), np.int32)
    img[:] = rng.integers(0, K)
    n_rect = max(4, int(4.0 * H * W / (mean_run * mean_run)))
    hs = np.maximum(2, rng.exponential(mean_run, n_rect)).astype(np.int64)
    ws = np.maximum(2, rng.exponential(mean_run, n_rect)).astype(np.int64)
    xs = rng.integers(-mean_run // 2, H, n_rect)
    ys = rng.integers(-mean_run // 2, W, n_rect)
    ls = rng.integers(0, K, n_rect)
    for h, w, x, y, l in zip(hs, ws, xs, ys, ls):
        img[max(0, x):max(0, x + h), max(0, y):max(0, y + w)] = l
    return img


def make_block(seed, H, W, S, K, diagonal, num_neighbor=8, noise=1.0, params=None, mean_run=25):
    """Returns dict(X[n,S] f64, edges[E,3] f64 (id1,id2,d), labels_true[n], params[K,3B+2],
    means[K,S], covars[K,S,S] (with the EM-time 2e-3 jitter), H, W, diagonal)."""
    rng = np.random.default_rng(seed)
    tt = R.TreeTables(tree_for(S))
    if params is None:
        params = sample_ou_params(rng, tt, K)
    means, covars = R.ou_params_to_means_covars(tt, params)       # carries +1e-3 I  (:1034)
    covars = covars + 1e-3 * np.eye(S)                            # EM-time covariances carry 2e-3 (:1522-1524)
    img = label_image(rng, H, W, K, mean_run)
    if diagonal:
        assert H == W
        ii, jj = np.triu_indices(H)
        lab = img[ii, jj]
    else:
        lab = img.reshape(-1)
    n = lab.shape[0]
    Lc = np.linalg.cholesky(covars)
    z = rng.standard_normal((n, S))
    X = means[lab] + noise * np.einsum("nij,nj->ni", Lc[lab], z)
    X = np.maximum(X, 0.0)
    edges = R.grid_edges(X, H, W, diagonal, num_neighbor)
    return dict(X=X, edges=edges, labels_true=lab.astype(np.int32), params=params, means=means,
                covars=covars, H=H, W=W, diagonal=bool(diagonal), tree=tree_for(S), num_neighbor=num_neighbor)


def make_knn_block(seed, n, S, K, k=6, noise=1.0, cells=12):
    """A labelling problem on a graph that is NOT the contact-map stencil (round 6: the general-graph boundary of
    pygco.cut_general_graph, phylo_hmrf.py:496-498): n points uniform in the unit square, each tied to its k nearest
    neighbours (undirected, id1 < id2, sorted like utility.py:1955-1960); ground-truth states constant on the cells of a
    cells x cells checkerboard of random states; x_i ~ N(mu_{l_i}, Sigma_{l_i}) clipped at 0; edge distance
    d_ij = |x_i - x_j|^2 / (|x_i| |x_j| + 1e-16) as utility.py:1935-1939.  Returns the dict of make_block (H = W = 0)."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    tt = R.TreeTables(tree_for(S))
    params = sample_ou_params(rng, tt, K)
    means, covars = R.ou_params_to_means_covars(tt, params)
    covars = covars + 1e-3 * np.eye(S)
    pts = rng.random((n, 2))
    board = rng.integers(0, K, (cells, cells))
    lab = board[np.minimum((pts[:, 0] * cells).astype(np.int64), cells - 1), np.minimum((pts[:, 1] * cells).astype(np.int64), cells - 1)]
    Lc = np.linalg.cholesky(covars)
    z = rng.standard_normal((n, S))
    X = np.maximum(means[lab] + noise * np.einsum("nij,nj->ni", Lc[lab], z), 0.0)
    _, idx = cKDTree(pts).query(pts, k=k + 1)
    a = np.repeat(np.arange(n), k)
    b = idx[:, 1:].reshape(-1)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    pair = np.unique(lo * np.int64(n) + hi)
    id1, id2 = pair // n, pair % n
    d = np.sum((X[id1] - X[id2]) ** 2, axis=1) / (np.linalg.norm(X[id1], axis=1) * np.linalg.norm(X[id2], axis=1) + 1e-16)
    edges = np.stack([id1.astype(np.float64), id2.astype(np.float64), d], axis=1)
    return dict(X=X, edges=edges, labels_true=lab.astype(np.int32), params=params, means=means, covars=covars, H=0, W=0,
                diagonal=False, tree=tree_for(S), points=pts)
