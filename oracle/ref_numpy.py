"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

float64 NumPy restatement of the reference's E-step hot path, one
function per reference item, each citing the reference file:line it follows.  Pinned by
``tests/test_oracle_golden.py`` against fixtures generated from the reference's OWN
functions (imported in the build container by ``tests/golden/make_golden.py``) and, for
the emission density, against ``scipy.stats.multivariate_normal.logpdf``.

Third-party arithmetic that is not under /root/reference:
  * scikit-learn 0.18 ``log_multivariate_normal_density(..., 'full')`` (README.md:80;
    call site phylo_hmrf.py:266-268; removed upstream in 0.20): restated in
    ``log_multivariate_normal_density_full`` from its published algorithm
    (Cholesky + triangular solve per state, +1e-7*I retry).  No reference test pins it,
    so it is pinned against scipy's logpdf instead (rtol 1e-10).
"""
import numpy as np
import scipy.linalg

SMALL_EPS = 1e-16  # phylo_hmrf.py:49


# --------------------------------------------------------------------------------------
# L0 tree tables  (phylo_hmrf.py:715-919)
# --------------------------------------------------------------------------------------
class TreeTables(object):
    """node_num, leaf_vec, parent_list, pair_list, A2, leaf_list of a rooted tree.

    edge_list: rows (a, b); the smaller index is the parent (phylo_hmrf.py:718-720).
    """

    def __init__(self, edge_list):
        edge_list = np.asarray(edge_list, dtype=np.int64)
        self.node_num = int(edge_list.max()) + 1            # :716
        n2 = self.node_num
        tree = np.zeros((n2, n2))
        for a, b in edge_list:                               # :718-720
            tree[min(a, b), max(a, b)] = 1
        self.tree_mtx = tree
        self.branch_dim = n2 - 1                             # :105
        self.n_params = n2 + 2 * self.branch_dim + 1         # :107
        # leaves = nodes without children                    # :855-865
        self.leaf_vec = np.array([i for i in range(n2) if not np.any(tree[i] > 0)], dtype=np.int64)
        # parent of every node                               # :883-888
        self.parent_list = [None] * n2
        for i in range(n2):
            b = np.where(tree[:, i] > 0)[0]
            self.parent_list[i] = int(b[0]) if b.shape[0] > 0 else []
        # root->leaf paths                                   # :837-853
        self.path_vec = []
        for leaf in self.leaf_vec:
            path = [int(leaf)]
            p = self.parent_list[leaf]
            while p != []:
                path.insert(0, p)
                p = self.parent_list[p]
            self.path_vec.append(np.array(path, dtype=np.int64))
        # reachable leaves per node -> leaf_list (node -> feature column)   # :728-769
        reach = [None] * n2

        def sub(i):
            kids = np.where(tree[i] == 1)[0]
            r = [i] if kids.shape[0] == 0 else [x for j in kids for x in sub(j)]
            reach[i] = r
            return r

        sub(0)
        self.branch_vec = reach
        self.leaf_list = {}
        cnt = 0
        for i in range(n2):
            if len(reach[i]) == 1:                           # :758-760
                self.leaf_list[i] = cnt
                cnt += 1
        # pair_list and A2                                   # :894-911
        n1 = len(self.leaf_vec)
        N1 = n1 * (n1 - 1) // 2
        self.A2 = np.zeros((N1, n2))
        self.pair_list = []
        cnt = 0
        for i in range(n1):
            v1 = self.path_vec[i]
            for j in range(i + 1, n1):
                v2 = self.path_vec[j]
                common = np.intersect1d(v1, v2)
                mrca = int(np.max(common))                   # :902
                self.A2[cnt, np.setdiff1d(v1, common)] = 1   # :904-907
                self.A2[cnt, np.setdiff1d(v2, common)] = 1
                self.pair_list.append([int(self.leaf_vec[i]), int(self.leaf_vec[j]), mrca])
                cnt += 1
        self.n_leaves = n1


# --------------------------------------------------------------------------------------
# a2  OU tree recursion -> (means_, _covars_)   (phylo_hmrf.py:985-1036)
# --------------------------------------------------------------------------------------
def ou_param_to_mean_cov(tt, params, min_covar=1e-3):
    """One state.  params = [root var | beta_1..B | lambda_1..B | theta_0..B]  (:995-997).

    Returns (mean[S], cov[S,S] INCLUDING +min_covar*I as at :1034, values[node_num,2]).
    """
    params = np.asarray(params, dtype=np.float64)
    n1, n2, B = tt.n_leaves, tt.node_num, tt.branch_dim
    p1 = params[1:]
    beta1, lambda1, theta1 = p1[0:B], p1[B:2 * B], p1[2 * B:3 * B + 1]
    ratio1 = np.zeros(B)
    b = np.where(beta1 > 1e-07)[0]                           # :999-1001
    ratio1[b] = lambda1[b] / (2 * beta1[b])
    values = np.zeros((n2, 2))
    values[0, 0] = theta1[0]                                 # :1002
    values[0, 1] = params[0]                                 # :1003
    bexp = np.insert(np.exp(-beta1), 0, 0)                   # :1004-1005
    beta1 = np.insert(beta1, 0, 0)                           # :1011
    ratio1 = np.insert(ratio1, 0, 0)
    p = tt.parent_list
    for i in range(1, n2):                                   # :1013-1015
        values[i, 0] = values[p[i], 0] * bexp[i] + theta1[i] * (1 - bexp[i])
        values[i, 1] = ratio1[i] * (1 - bexp[i] ** 2) + values[p[i], 1] * (bexp[i] ** 2)
    pair = np.array(tt.pair_list)
    s1 = np.matmul(tt.A2, beta1)                             # :1018
    s2 = values[pair[:, -1], 1] * np.exp(-s1)                # :1019-1020
    cov = np.zeros((n1, n1))
    for k in range(pair.shape[0]):                           # :1024-1028
        i, j = tt.leaf_list[pair[k, 0]], tt.leaf_list[pair[k, 1]]
        cov[i, j] = s2[k]
        cov[j, i] = s2[k]
    for i in range(n1):                                      # :1030-1031
        cov[i, i] = values[tt.leaf_vec[i], 1]
    mean = values[tt.leaf_vec, 0].copy()                     # :1033
    return mean, cov + min_covar * np.eye(n1), values        # :1034


def ou_params_to_means_covars(tt, params_vec, min_covar=1e-3):
    """All K states: `_ou_param_varied_constraint` (phylo_hmrf.py:985-1036)."""
    K = params_vec.shape[0]
    S = tt.n_leaves
    means, covars = np.zeros((K, S)), np.zeros((K, S, S))
    for c in range(K):
        means[c], covars[c], _ = ou_param_to_mean_cov(tt, params_vec[c], min_covar)
    return means, covars


# --------------------------------------------------------------------------------------
# a1  emission  (phylo_hmrf.py:266-268 -> sklearn 0.18 log_multivariate_normal_density 'full')
# --------------------------------------------------------------------------------------
def log_multivariate_normal_density_full(X, means, covars, min_covar=1e-7):
    X = np.asarray(X, dtype=np.float64)
    n, S = X.shape
    K = len(means)
    out = np.empty((n, K))
    for c in range(K):
        mu, cv = means[c], covars[c]
        try:
            L = scipy.linalg.cholesky(cv, lower=True)
        except scipy.linalg.LinAlgError:
            try:
                L = scipy.linalg.cholesky(cv + min_covar * np.eye(S), lower=True)
            except scipy.linalg.LinAlgError:
                raise ValueError("'covars' must be symmetric, positive-definite")
        logdet = 2 * np.sum(np.log(np.diagonal(L)))
        sol = scipy.linalg.solve_triangular(L, (X - mu).T, lower=True).T
        out[:, c] = -0.5 * (np.sum(sol ** 2, axis=1) + S * np.log(2 * np.pi) + logdet)
    return out


# --------------------------------------------------------------------------------------
# graph helpers  (phylo_hmrf.py:524-536, :567-598, :674-689)
# --------------------------------------------------------------------------------------
def potts_matrix(K, beta):
    """`_pairwise_potential` (phylo_hmrf.py:524-536): beta*(1-delta)."""
    V = np.full((K, K), float(beta))
    np.fill_diagonal(V, 0.0)
    return V


def edge_weights_from_distance(edge_list, beta1):
    """phylo_hmrf.py:585,589: w = exp(-beta1*d), ids = int64(edge_list[:, 0:2])."""
    edge_list = np.asarray(edge_list)
    return np.exp(-beta1 * edge_list[:, 2]), np.int64(edge_list[:, 0:2])


def connected_edge(edge_ids, n):
    """`_connected_edge` (phylo_hmrf.py:674-689): per-node list of incident edge ids."""
    inc = [[] for _ in range(n)]
    for e, (j, i) in enumerate(edge_ids):
        inc[i].append(e)
        inc[j].append(e)
    return inc


# --------------------------------------------------------------------------------------
# a5  pairwise potential  (phylo_hmrf.py:398-436), vectorised, same sums
# --------------------------------------------------------------------------------------
def pairwise_compare(label, edge_ids, w, V, estimate_type):
    """pp[i,k] = sum_{e in inc(i)} V[l_other(e), k] * (w_e if estimate_type==3 else 1);
    isolated node -> V[l_i, :] (:421-423)."""
    label = np.asarray(label).astype(np.int64)
    n, K = label.shape[0], V.shape[0]
    pp = np.zeros((n, K))
    a, b = edge_ids[:, 0], edge_ids[:, 1]
    ww = w if estimate_type == 3 else np.ones_like(w)
    np.add.at(pp, a, V[label[b]] * ww[:, None])
    np.add.at(pp, b, V[label[a]] * ww[:, None])
    deg = np.bincount(np.concatenate([a, b]), minlength=n)
    iso = np.where(deg == 0)[0]
    pp[iso] = V[label[iso]]
    return pp


def pairwise_compare_loops(label, edge_ids, w, V, estimate_type):
    """Literal per-node / per-edge loops of phylo_hmrf.py:398-436 (ref-faithful timing variant)."""
    n, K = len(label), V.shape[0]
    inc = connected_edge(edge_ids, n)
    out = []
    for i in range(n):
        idx = inc[i]
        if len(idx) == 0:
            out.append(V[int(label[i])])
            continue
        ep = np.zeros(K)
        for k in idx:
            id1 = edge_ids[k]
            k1 = id1[id1 != i][0]
            s = int(label[k1])
            ep = ep + (V[s] * w[k] if estimate_type == 3 else V[s])
        out.append(ep)
    return np.asarray(out)


# --------------------------------------------------------------------------------------
# a6 + a7  posteriors and the four cost scalars  (phylo_hmrf.py:334-396, :438-468)
# --------------------------------------------------------------------------------------
def compute_posteriors_graph(label, logprob, edge_ids, w, V, estimate_type):
    """Returns (posteriors, pairwise_cost, pairwise_cost_normalize, unary_cost, cost1)
    exactly as `_compute_posteriors_graph` (phylo_hmrf.py:334-355)."""
    label = np.asarray(label).astype(np.int64)
    n, K = logprob.shape
    pp = pairwise_compare(label, edge_ids, w, V, estimate_type)
    wp = np.exp(logprob - pp)                                # :342 (no max shift)
    post = wp / wp.sum(axis=1, keepdims=True)                # :343-345
    pr = np.exp(-pp)                                         # :347
    ppn = pr / pr.sum(axis=1, keepdims=True)                 # :348-350
    # _pairwise_compare_ensemble/_single (:438-468): per node, setdiff1d of incident edge
    # endpoints minus i (unique, ascending neighbour ids) paired POSITIONALLY with the
    # incident-edge weights in edge order.  For a simple graph whose incident edges are
    # already ordered by neighbour id this is sum_e V[l_other, l_i] * w_e.
    pc = pairwise_cost_ensemble(label, edge_ids, w, V, estimate_type)
    idx = np.arange(n)
    unary_cost = -np.sum(logprob[idx, label]) * 1.0 / n      # :385-390
    pcn = -np.sum(np.log(ppn[idx, label] + SMALL_EPS)) * 1.0 / n   # :386,392
    return post, pc, pcn, unary_cost, unary_cost + pcn       # :394


def pairwise_cost_ensemble(label, edge_ids, w, V, estimate_type):
    """(1/n) sum_i sum_{e in inc(i)} V[l_other, l_i] * w_e  (phylo_hmrf.py:438-447);
    every edge counted from both endpoints.  Valid for simple graphs (no parallel edges),
    which is what utility.py:1955-1960 builds."""
    label = np.asarray(label).astype(np.int64)
    n = label.shape[0]
    a, b = edge_ids[:, 0], edge_ids[:, 1]
    ww = w if estimate_type == 3 else np.ones_like(w)
    tot = np.sum(V[label[b], label[a]] * ww) + np.sum(V[label[a], label[b]] * ww)
    return tot * 1.0 / n


def pairwise_cost_ensemble_loops(label, edge_ids, w, V, estimate_type):
    """Literal loop form of phylo_hmrf.py:438-468 including the setdiff1d quirk."""
    n = len(label)
    inc = connected_edge(edge_ids, n)
    label = np.asarray(label).astype(np.int64)
    cost = np.zeros(n)
    for i in range(n):
        t_idx = inc[i]
        if len(t_idx) == 0:
            continue
        temp1 = edge_ids[t_idx]
        id1 = np.setdiff1d(temp1.ravel(), i)
        ep = V[label[id1], label[i]]
        if estimate_type == 3:
            ep = ep * w[t_idx]
        cost[i] = sum(ep)
    return np.sum(cost) * 1.0 / n


# --------------------------------------------------------------------------------------
# a8  sufficient statistics  (phylo_hmrf.py:311-314)
# --------------------------------------------------------------------------------------
def sufficient_statistics(posteriors, X):
    return {
        "post": posteriors.sum(axis=0),
        "obs": np.dot(posteriors.T, X),
        "obs*obs.T": np.einsum("ij,ik,il->jkl", posteriors, X, X),
    }


# --------------------------------------------------------------------------------------
# a-E  judged energy
# --------------------------------------------------------------------------------------
def mrf_energy(label, logprob, edge_ids, w, beta):
    """E_float = sum_i -logprob[i,l_i] + beta * sum_(i,j) w_ij [l_i != l_j]."""
    label = np.asarray(label).astype(np.int64)
    n = label.shape[0]
    e_un = -np.sum(logprob[np.arange(n), label])
    e_pw = beta * np.sum(w * (label[edge_ids[:, 0]] != label[edge_ids[:, 1]]))
    return float(e_un + e_pw), float(e_un), float(e_pw)


# --------------------------------------------------------------------------------------
# grid edge builder (utility.py:1871-2053), vectorised restatement used by the synthetic
# generator and by tests (the product has its own copy in phylo_hmrf_amd.graph)
# --------------------------------------------------------------------------------------
def grid_edges(X, H, W, diagonal, num_neighbor=8):
    """Edge list [E,3] = (id1, id2, d) sorted by (id1, id2).

    diagonal=True : N x N block stored upper-triangular row-major (utility.py:2310-2317),
                    half-stencil restricted to row<=col (utility.py:1917-1920), diag-diag
                    edge distance halved (:1942-1953).
    diagonal=False: full H x W block (utility.py:1975-2053).
    d = |x1-x2|^2 / (|x1||x2| + 1e-16)  (utility.py:1935-1939).
    """
    if num_neighbor == 8:
        offs = [(0, 1), (1, 1), (1, 0), (1, -1)]             # utility.py:1899-1902
    else:
        offs = [(0, 1), (1, 0)]                              # :1904-1905
    if diagonal:
        N = H
        ii, jj = np.triu_indices(N)
        idmap = -np.ones((N, N), dtype=np.int64)
        idmap[ii, jj] = np.arange(ii.shape[0])
    else:
        ii, jj = np.divmod(np.arange(H * W), W)
        idmap = np.arange(H * W).reshape(H, W)
    norm = np.sqrt(np.sum(X * X, axis=1))
    out = []
    for dx, dy in offs:
        x2, y2 = ii + dx, jj + dy
        if diagonal:
            ok = (x2 <= y2) & (x2 >= 0) & (y2 < H)           # utility.py:1917-1918
        else:
            ok = (x2 >= 0) & (x2 < H) & (y2 >= 0) & (y2 < W)  # :2011
        id1 = np.where(ok)[0]
        id2 = idmap[x2[id1], y2[id1]]
        d = np.sum((X[id1] - X[id2]) ** 2, 1) / (norm[id1] * norm[id2] + 1e-16)
        if diagonal:
            dd = (ii[id1] == jj[id1]) & (x2[id1] == y2[id1])  # :1942-1953
            d = np.where(dd, 0.5 * d, d)
        out.append(np.stack([id1.astype(np.float64), id2.astype(np.float64), d], axis=1))
    e = np.concatenate(out, axis=0)
    order = np.lexsort((e[:, 1], e[:, 0]))
    return e[order]


def kmeans_step(X, centers):
    """One Lloyd step (model of csrc/init.hip; the reference's own initialiser is sklearn MiniBatchKMeans,
    phylo_hmrf.py:234-238, third-party and unpinned): nearest centre by squared Euclidean distance, lowest index on
    ties -> (labels[n], sums[K,S], counts[K], inertia)."""
    X = np.asarray(X, dtype=np.float64)
    centers = np.asarray(centers, dtype=np.float64)
    d2 = ((X[:, None, :] - centers[None, :, :]) ** 2).sum(axis=2)
    lab = np.argmin(d2, axis=1)
    K = centers.shape[0]
    sums = np.zeros_like(centers)
    np.add.at(sums, lab, X)
    return lab, sums, np.bincount(lab, minlength=K).astype(np.float64), float(d2[np.arange(X.shape[0]), lab].sum())


# --------------------------------------------------------------------------------------
# filter_mode 1 (utility.py:1575-1582): skimage.restoration.denoise_bilateral(img, sigma_color, sigma_spatial,
# multichannel=False).  scikit-image is a third-party dependency that is neither installed here nor vendored under
# /root/reference (the reference's README pins no version; its Python-2 era means 0.13 / 0.14): PARITY UNPINNED.  This
# restates the published algorithm of `skimage/restoration/_denoise_cy.pyx::_denoise_bilateral` for one channel, loop by
# loop over the window offsets (vectorised over the pixels): zero padding (mode 'constant', cval 0), spatial table
# exp(-0.5 (d / sigma_spatial)^2), colour table of `bins` entries exp(-0.5 (b max / bins / sigma_color)^2) indexed by
# min(int(|centre - value| * bins / max), bins - 1).
# --------------------------------------------------------------------------------------
def denoise_bilateral(img, sigma_color=None, sigma_spatial=1, win_size=None, bins=10000):
    image = np.asarray(img, dtype=np.float64)
    if win_size is None:
        win_size = max(5, 2 * int(np.ceil(3 * sigma_spatial)) + 1)
    sigma_color = sigma_color or image.std()
    mn, mx = image.min(), image.max()
    if mn == mx:
        return image.copy()
    if mn < 0.0:
        raise ValueError("Image must contain only positive values")
    ext = (win_size - 1) // 2
    color_lut = np.exp(-0.5 * (np.arange(bins) * mx / bins / sigma_color) ** 2)
    dist_scale = bins / mx
    rows, cols = image.shape
    padded = np.zeros((rows + 2 * ext, cols + 2 * ext))
    padded[ext:ext + rows, ext:ext + cols] = image
    total = np.zeros_like(image)
    weight = np.zeros_like(image)
    for wr in range(-ext, ext + 1):
        for wc in range(-ext, ext + 1):
            value = padded[ext + wr:ext + wr + rows, ext + wc:ext + wc + cols]
            range_w = np.exp(-0.5 * (np.sqrt(float(wr * wr + wc * wc)) / sigma_spatial) ** 2)
            b = np.minimum((np.abs(image - value) * dist_scale).astype(np.int64), bins - 1)
            w = range_w * color_lut[b]
            total += value * w
            weight += w
    return total / weight
