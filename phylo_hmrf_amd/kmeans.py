"""k-means initialisation on the device-resident observations.

The reference clusters the observations with sklearn's MiniBatchKMeans (phylo_hmrf.py:234-238) to get the initial state
labels and means.  Here the assignment + accumulation step runs on the GPU where X already lives
(`phmrf_kmeans_step`, csrc/init.hip); this module is the Lloyd loop around it: k-means++ seeding on a host sample,
per-iteration reduction of K(S+1)+1 doubles over blocks (and ranks), a few restarts, best inertia wins.
It is an alternative initialiser, not a restatement of MiniBatchKMeans (whose result depends on sklearn's version and
RNG stream): parity of the EM path does not depend on it.

`minibatch_centers` + `device_moments` are the default initialiser (init_method "minibatch"): the reference's own
estimator with the reference's settings finds the centres -- on all rows when there are at most `sample_cap` of them (then
these are exactly the reference's centres), else on a uniform host sample of that many rows (mini-batch k-means only ever
touches random batches of 2000 rows) -- and everything that is a pass over ALL nodes runs on the device: the assignment
of every node to its nearest centre, the per-cluster sums and second moments for the per-cluster OU fit, the global
covariance.
"""
import numpy as np


def kmeanspp_seeds(sample, K, rng):
    """k-means++ (Arthur & Vassilvitskii 2007) on a host sample [m,S] -> centres [K,S]."""
    sample = np.asarray(sample, dtype=np.float64)
    m = sample.shape[0]
    centers = np.empty((K, sample.shape[1]))
    centers[0] = sample[rng.integers(m)]
    d2 = np.sum((sample - centers[0]) ** 2, axis=1)
    for k in range(1, K):
        tot = d2.sum()
        idx = rng.integers(m) if not (tot > 0) else int(np.searchsorted(np.cumsum(d2), rng.random() * tot))
        centers[k] = sample[min(idx, m - 1)]
        d2 = np.minimum(d2, np.sum((sample - centers[k]) ** 2, axis=1))
    return centers


def _step(blocks, centers, reducer, write_labels=False):
    K, S = centers.shape
    acc = np.zeros(K * S + K + 1)
    for b in blocks:
        sums, counts, inertia = b.kmeans_step(centers, write_labels)
        acc[:K * S] += sums.ravel()
        acc[K * S:K * S + K] += counts
        acc[-1] += inertia
    if reducer is not None:
        acc = reducer.allreduce(acc)
    return acc[:K * S].reshape(K, S), acc[K * S:K * S + K], float(acc[-1])


def device_kmeans(blocks, sample, K, rng, reducer=None, n_init=3, max_iter=100, tol=1e-4):
    """Lloyd iterations over all blocks.  -> (centres [K,S], inertia).  The winning assignment is left in the blocks'
    labels.  `sample`: host rows used for seeding and for re-seeding empty clusters (identical on every rank)."""
    sample = np.asarray(sample, dtype=np.float64)
    scale = float(np.mean(np.var(sample, axis=0))) + 1e-300
    best = None
    for _ in range(max(1, n_init)):
        centers = kmeanspp_seeds(sample, K, rng)
        inertia = np.inf
        for _it in range(max_iter):
            sums, counts, inertia = _step(blocks, centers, reducer)
            new = centers.copy()
            live = counts > 0
            new[live] = sums[live] / counts[live][:, None]
            for k in np.flatnonzero(~live):                       # empty cluster: restart it at a random sample row
                new[k] = sample[rng.integers(sample.shape[0])]
            shift = float(np.sum((new - centers) ** 2))
            centers = new
            if shift <= tol * scale and live.all():
                break
        if best is None or inertia < best[1]:
            best = (centers.copy(), inertia)
    centers = best[0]
    _, _, inertia = _step(blocks, centers, reducer, write_labels=True)
    return centers, inertia


def minibatch_centers(X, K, seed, sample_cap=2000000):
    """The reference's clustering call (phylo_hmrf.py:234-238: MiniBatchKMeans, batch 2000, max_iter 1000, n_init 10) on
    the rows of X, or on a uniform sample of `sample_cap` rows when there are more.  -> centres [K,S]."""
    from sklearn import cluster
    X = np.asarray(X)
    n = X.shape[0]
    if n > sample_cap:
        rows = np.sort(np.random.default_rng(seed).choice(n, size=sample_cap, replace=False))
        X = X[rows]
    km = cluster.MiniBatchKMeans(n_clusters=K, random_state=seed, batch_size=2000, max_iter=1000, n_init=10)
    km.fit(X)
    return np.asarray(km.cluster_centers_, dtype=np.float64)


def device_moments(blocks, centers, reducer=None, write_labels=True):
    """Every node to its nearest centre (the assignment becomes the blocks' labels) and the per-cluster statistics of the
    assignment, on the device: -> (counts[K], sums[K,S], outer[K,S,S], inertia), summed over blocks and ranks."""
    K, S = centers.shape
    acc = np.zeros(K * S + K + 1 + K * S * S)
    for b in blocks:
        sums, counts, inertia, outer = b.kmeans_moments(centers, write_labels)
        acc[:K * S] += sums.ravel()
        acc[K * S:K * S + K] += counts
        acc[K * S + K] += inertia
        acc[K * S + K + 1:] += outer.ravel()
    if reducer is not None:
        acc = reducer.allreduce(acc)
    return (acc[K * S:K * S + K], acc[:K * S].reshape(K, S), acc[K * S + K + 1:].reshape(K, S, S), float(acc[K * S + K]))

