"""Species-tree tables and the Ornstein-Uhlenbeck parameter -> (mean, covariance) map, vectorised over states.

Host side (tiny, O(K) per EM iteration).  Semantics follow the reference:
  tree tables   phylo_hmrf.py:715-919  (_initilize_tree_mtx, _search_leaf, _search_ancestor, _matrix1,
                                        _compute_base_struct's leaf -> feature column map)
  OU recursion  phylo_hmrf.py:985-1036 (_ou_param_varied_constraint)
Parameter vector of one state (phylo_hmrf.py:995-997, n_params = 3B+2, :107):
  [ root variance | beta_1..beta_B | lambda_1..lambda_B | theta_0..theta_B ]      (B = number of branches)
"""
import numpy as np


class PhyloTree(object):
    def __init__(self, edge_list):
        e = np.asarray(edge_list, dtype=np.int64).reshape(-1, 2)
        self.node_num = int(e.max()) + 1
        N = self.node_num
        self.branch_dim = N - 1
        self.n_params = N + 2 * self.branch_dim + 1
        # the smaller index of an edge is the parent (:718-720); a node's parent is its smallest-index one (:884-886)
        parent = np.full(N, -1, dtype=np.int64)
        children = [[] for _ in range(N)]
        for a, b in e:
            p, c = (a, b) if a < b else (b, a)
            children[p].append(int(c))
            if parent[c] < 0 or p < parent[c]:
                parent[c] = p
        self.parent = parent
        self.leaf_vec = np.array([i for i in range(N) if not children[i]], dtype=np.int64)   # :855-865
        S = len(self.leaf_vec)
        self.n_leaves = S
        # root -> node paths
        self.paths = []
        for leaf in self.leaf_vec:
            path, p = [int(leaf)], parent[leaf]
            while p >= 0:
                path.append(int(p))
                p = parent[p]
            self.paths.append(path[::-1])
        # leaf -> feature column: nodes that reach exactly one leaf, numbered in node order (:755-760)
        n_reach = np.zeros(N, dtype=np.int64)
        for path in self.paths:
            n_reach[path] += 1
        col, cnt = {}, 0
        for i in range(N):
            if n_reach[i] == 1:
                col[i] = cnt
                cnt += 1
        self.leaf_col = np.array([col[int(l)] for l in self.leaf_vec], dtype=np.int64)
        self.n_features = S
        if cnt != S or not np.array_equal(self.leaf_col, np.arange(S)):
            # the reference mixes two leaf orders (values[leaf_vec] for means/diagonal, leaf_list for the
            # off-diagonal) which only agree when every node that reaches a single leaf IS that leaf
            raise ValueError("unsupported tree: an internal node has a single leaf below it")
        # leaf pairs: nearest common ancestor and the branches between the two leaves (:894-911)
        pa, pb, anc, path_mask = [], [], [], []
        for i in range(S):
            si = set(self.paths[i])
            for j in range(i + 1, S):
                common = si & set(self.paths[j])
                mrca = max(common)                                                      # :902
                mask = np.zeros(N)
                for v in self.paths[i] + self.paths[j]:
                    if v not in common:
                        mask[v] = 1.0
                pa.append(i)
                pb.append(j)
                anc.append(mrca)
                path_mask.append(mask)
        self.pair_a = np.array(pa, dtype=np.int64)          # indices into leaf_vec
        self.pair_b = np.array(pb, dtype=np.int64)
        self.pair_anc = np.array(anc, dtype=np.int64)
        self.A2 = np.array(path_mask).reshape(len(pa), N)   # [pairs, node_num]
        # topological order root-first (parents have smaller indices in the reference's trees, but do not rely on it)
        order, seen = [], np.zeros(N, dtype=bool)
        stack = [i for i in range(N) if parent[i] < 0]
        while stack:
            v = stack.pop()
            if seen[v]:
                continue
            seen[v] = True
            order.append(v)
            stack.extend(children[v])
        self.order = [v for v in order if parent[v] >= 0]

    # -- parameter slices --------------------------------------------------------------------------
    def split(self, params):
        B = self.branch_dim
        p = np.asarray(params, dtype=np.float64)
        return p[..., 0], p[..., 1:1 + B], p[..., 1 + B:1 + 2 * B], p[..., 1 + 2 * B:2 + 3 * B]

    def node_moments(self, params):
        """params [..., 3B+2] -> (mean[..., N], var[..., N], e[..., N], ratio[..., N]) of every tree node."""
        v0, beta, lam, theta = self.split(params)
        shp = beta.shape[:-1]
        N = self.node_num
        e = np.concatenate([np.zeros(shp + (1,)), np.exp(-beta)], axis=-1)               # :1004-1005
        ok = beta > 1e-07                                                                 # :999-1001
        ratio = np.where(ok, lam / (2.0 * np.where(ok, beta, 1.0)), 0.0)
        ratio = np.concatenate([np.zeros(shp + (1,)), ratio], axis=-1)
        mean = np.zeros(shp + (N,))
        var = np.zeros(shp + (N,))
        roots = np.flatnonzero(self.parent < 0)
        mean[..., roots] = theta[..., roots]                                              # :1002
        var[..., roots] = v0[..., None]                                                   # :1003
        for i in self.order:                                                              # :1013-1015
            p = self.parent[i]
            mean[..., i] = mean[..., p] * e[..., i] + theta[..., i] * (1.0 - e[..., i])
            var[..., i] = ratio[..., i] * (1.0 - e[..., i] ** 2) + var[..., p] * e[..., i] ** 2
        return mean, var, e, ratio

    def mean_cov(self, params, min_covar=1e-3):
        """`_ou_param_varied_constraint` for a batch: params [..., 3B+2] -> means [..., S], covars [..., S, S]
        (covars include + min_covar * I as at phylo_hmrf.py:1034)."""
        params = np.asarray(params, dtype=np.float64)
        _, beta, _, _ = self.split(params)
        shp = beta.shape[:-1]
        mean, var, _, _ = self.node_moments(params)
        S = self.n_features
        beta_full = np.concatenate([np.zeros(shp + (1,)), beta], axis=-1)
        s1 = np.einsum("pn,...n->...p", self.A2, beta_full)                               # :1018
        s2 = var[..., self.pair_anc] * np.exp(-s1)                                        # :1019-1020
        cov = np.zeros(shp + (S, S))
        ca, cb = self.leaf_col[self.pair_a], self.leaf_col[self.pair_b]
        cov[..., ca, cb] = s2
        cov[..., cb, ca] = s2
        cov[..., self.leaf_col, self.leaf_col] = var[..., self.leaf_vec]                  # :1030-1031
        means = np.zeros(shp + (S,))
        means[..., self.leaf_col] = mean[..., self.leaf_vec]                              # :1033
        # the reference indexes means by leaf order (values[leaf_vec]); leaf_col is the identity (checked in __init__)
        return means, cov + min_covar * np.eye(S)


def load_tree_files(data_path):
    """edge.1.txt / branch_length.1.txt / species_name.1.txt as read by run() (phylo_hmrf.py:1607-1631)."""
    import os
    edge_list = [list(map(int, line.split("\t"))) for line in open(os.path.join(data_path, "edge.1.txt")) if line.strip()]
    branch_list = None
    fn = os.path.join(data_path, "branch_length.1.txt")
    if os.path.exists(fn):
        rows = [list(map(float, line.split("\t"))) for line in open(fn) if line.strip()]
        branch_list = rows[0]
    species = None
    fn = os.path.join(data_path, "species_name.1.txt")
    if os.path.exists(fn):
        species = [line.strip() for line in open(fn) if line.strip()]
    return edge_list, branch_list, species
