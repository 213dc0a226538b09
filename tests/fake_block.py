"""TEST DOUBLE (tests only): the `Block` interface backed by the CPU oracle, so the host logic of the EM driver
(bookkeeping, sharding, reduction) can be exercised without a GPU.  With `labeller="gco"` the labelling step is the
reference's own gco swap through pygco's quantisation, which makes the driver reproducible against the EM trace
recorded from the reference (tests/golden/em_trace.npz)."""
import numpy as np

from oracle import gco_ref, mrf_moves, ref_numpy as R


class FakeBlock(object):
    labeller = "gco"

    def __init__(self, n, S, K):
        self.n, self.S, self.K = n, S, K
        self.slots = {}
        self.labels = np.zeros(n, dtype=np.int64)
        self.grid = None

    def close(self):
        pass

    def sync(self):
        pass

    def set_observations(self, X):
        self.X = np.asarray(X, dtype=np.float64)

    def set_graph(self, edges, w):
        self.eid = np.int64(edges)
        self.w = np.asarray(w, dtype=np.float64)

    def set_grid(self, H, W, diagonal, num_neighbor=8):
        self.grid = (H, W, bool(diagonal), num_neighbor)

    def set_labels(self, labels):
        self.labels = np.asarray(labels).astype(np.int64)

    def kmeans_step(self, centers, write_labels=False):
        from oracle import ref_numpy as R
        lab, sums, counts, inertia = R.kmeans_step(self.X, centers)
        if write_labels:
            self.labels = lab.astype(np.int64)
        return sums, counts, inertia

    def kmeans_moments(self, centers, write_labels=False):
        from oracle import ref_numpy as R
        sums, counts, inertia = self.kmeans_step(centers, write_labels)
        lab, _, _, _ = R.kmeans_step(self.X, centers)
        K, S = centers.shape
        outer = np.zeros((K, S, S))
        np.add.at(outer, lab, self.X[:, :, None] * self.X[:, None, :])
        return sums, counts, inertia, outer

    def get_labels(self):
        return self.labels.astype(np.int32)

    def save_labels(self, slot):
        self.slots[slot] = self.labels.copy()

    def restore_labels(self, slot):
        self.labels = self.slots[slot].copy()

    def get_saved_labels(self, slot):
        return self.slots[slot].astype(np.int32)

    def warm_start(self, beta, slot, choose=True, report=True):
        ec = R.mrf_energy(np.int64(self.labels), self.logprob, self.eid, self.w, beta)[0]
        es = R.mrf_energy(np.int64(self.slots[slot]), self.logprob, self.eid, self.w, beta)[0]
        took = not (ec < es)
        if choose and took:
            self.restore_labels(slot)
        return ec, es, took

    def emission(self, means, covars):
        self.logprob = R.log_multivariate_normal_density_full(self.X, means, covars)

    def get_logprob(self):
        return self.logprob

    def set_logprob(self, lp):
        self.logprob = np.asarray(lp, dtype=np.float64)

    def solve_fast(self, beta, **kw):
        V = R.potts_matrix(self.K, beta)
        if self.labeller == "gco" and gco_ref.available():
            self.labels = gco_ref.cut_general_graph(self.eid, self.w, -self.logprob, V, n_iter=5000, algorithm="swap",
                                                    init_labels=self.labels).astype(np.int64)
        else:
            g = mrf_moves.Graph(self.n, self.eid, self.w)
            H, W, diag, nn = self.grid
            self.labels = mrf_moves.solve(g, -self.logprob, self.labels, beta, H, W, diag, nn)

    def posterior_stats(self, beta, estimate_type, want_posteriors=False):
        V = R.potts_matrix(self.K, beta)
        post, pc, pcn, uc, c1 = R.compute_posteriors_graph(self.labels, self.logprob, self.eid, self.w, V, estimate_type)
        st = R.sufficient_statistics(post, self.X)
        return st, np.array([pc, pcn, uc, c1]) * self.n, (post if want_posteriors else None)
