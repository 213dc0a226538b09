"""k-means initialisation on the device-resident observations.

The reference clusters the observations with sklearn's MiniBatchKMeans (phylo_hmrf.py:234-238) to get the initial state
labels and means.  Here the assignment + accumulation step runs on the GPU where X already lives
(`phmrf_kmeans_step`, csrc/init.hip); this module is the Lloyd loop around it: k-means++ seeding on a host sample,
per-iteration reduction of K(S+1)+1 doubles over blocks (and ranks), a few restarts, best inertia wins.
It is an alternative initialiser, not a restatement of MiniBatchKMeans (whose result depends on sklearn's version and
RNG stream): parity of the EM path does not depend on it.
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
