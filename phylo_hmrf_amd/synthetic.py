"""Seeded synthetic multi-species Hi-C blocks for the benchmark, generated on the device.

Ground-truth label image of random rectangles (mean run ~25 bins) over K states; x_i ~ N(mu_{l_i}, Sigma_{l_i})
clipped at 0 (the reference's features are log(1+x) >= 0, utility.py:363).  The neighbour graph is NOT built here:
phmrf_block_build_grid_graph derives it from X on the device.
"""
import numpy as np

TREE4 = [[0, 1], [1, 2], [1, 3], [3, 4], [4, 5], [4, 6], [3, 7]]           # the example tree's topology
TREE8 = [[0, 1], [0, 2], [1, 3], [1, 4], [2, 5], [2, 6], [3, 7], [3, 8], [4, 9], [4, 10],
         [5, 11], [5, 12], [6, 13], [6, 14]]                                # balanced, 8 leaves


def tree_for(S):
    if S == 4:
        return TREE4
    if S == 8:
        return TREE8
    raise ValueError("synthetic trees are defined for S in {4, 8}")


def sample_ou_params(rng, tree, K):
    """OU parameters in the reference's feasible box (bounds phylo_hmrf.py:1365-1366, :1412-1413)."""
    B = tree.branch_dim
    P = np.zeros((K, tree.n_params))
    P[:, 0] = rng.uniform(0.1, 1.0, K)
    P[:, 1:1 + B] = rng.uniform(0.05, 2.0, (K, B))
    P[:, 1 + B:1 + 2 * B] = rng.uniform(0.05, 2.0, (K, B))
    P[:, 1 + 2 * B:] = rng.uniform(0.0, 4.0, (K, B + 1))
    return P


def label_image(rng, H, W, K, mean_run=25):
    img = np.full((H, W), int(rng.integers(0, K)), dtype=np.int64)
    n_rect = max(4, int(4.0 * H * W / (mean_run * mean_run)))
    hs = np.maximum(2, rng.exponential(mean_run, n_rect)).astype(np.int64)
    ws = np.maximum(2, rng.exponential(mean_run, n_rect)).astype(np.int64)
    xs = rng.integers(-mean_run // 2, H, n_rect)
    ys = rng.integers(-mean_run // 2, W, n_rect)
    ls = rng.integers(0, K, n_rect)
    for h, w, x, y, l in zip(hs, ws, xs, ys, ls):
        img[max(0, x):max(0, x + h), max(0, y):max(0, y + w)] = l
    return img


def device_observations(torch, dev, seed, H, W, diagonal, K, means, covars, chunk=4 << 20, mean_run=25, noise=1.0):
    """-> torch f32 [n, S] on `dev` for one block (upper triangle row-major when diagonal).  mean_run: mean side of the
    ground truth's rectangles (bins); noise: factor on the states' standard deviations (1: as sampled)."""
    rng = np.random.default_rng(seed)
    img = label_image(rng, H, W, K, mean_run)
    if diagonal:
        ii, jj = np.triu_indices(H)
        lab = img[ii, jj]
        del ii, jj
    else:
        lab = img.reshape(-1)
    del img
    n, S = lab.shape[0], means.shape[1]
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    mu = torch.tensor(means, dtype=torch.float32, device=dev)
    L = torch.tensor(np.linalg.cholesky(covars), dtype=torch.float32, device=dev)
    out = torch.empty((n, S), dtype=torch.float32, device=dev)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        l = torch.from_numpy(lab[s:e]).to(dev)
        z = torch.randn((e - s, S), generator=gen, device=dev, dtype=torch.float32)
        x = mu[l] + float(noise) * torch.einsum("nij,nj->ni", L[l], z)
        out[s:e] = torch.clamp_min(x, 0.0)
    return out
