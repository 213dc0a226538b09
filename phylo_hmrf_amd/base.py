"""EM driver with the reference's fit surface (`_BaseGraph.fit_accumulate_test`, base.py:301-455).

What is kept: the method name, argument order, return tuple, the per-iteration bookkeeping (cost weighting by
n_r/N, cost_vec rows, min_cost / min_cost1 tracking, labels_local warm start, t_labels from iteration 3 on,
the three stopping rules, no M-step after the last E-step) and the per-iteration prints.
What is different: the E-step of every syntenic block runs on the GPU (no fork / Queue / pickle, base.py:352-372);
labels stay on the device (labels_local / t_labels are device snapshots) and only the K*(1+S+S*S)+4 numbers of the
reduction leave it; with several ranks the blocks are sharded and the reduction is one all-reduce.
"""
from __future__ import print_function

import time

import numpy as np

from .block import pack_stats, unpack_stats

SLOT_LOCAL = 0      # labels_local: warm start of the next labelling (base.py:419, phylo_hmrf.py:479)
SLOT_BEST3 = 1      # t_labels: labels at the best cost1 since iteration 3 (base.py:422-426)
SLOT_CURRENT = 2


class _BaseGraph(object):
    """Base class for Markov Random Field models (reference: base.py:96-594)."""

    def __init__(self, n_components=1, run_id=0, estimate_type=0, weight_type=0, startprob_prior=1.0,
                 transmat_prior=1.0, algorithm="viterbi", random_state=None, n_iter=10, tol=1e-2, verbose=False,
                 params="stmc", init_params="stmc"):
        self.n_components = n_components
        self.params = params
        self.init_params = init_params
        self.startprob_prior = startprob_prior
        self.transmat_prior = transmat_prior
        self.algorithm = algorithm
        self.random_state = random_state
        self.n_iter = n_iter
        self.tol = tol
        self.verbose = verbose
        self.run_id = run_id
        self.estimate_type = estimate_type
        self.weight_type = weight_type
        self.quiet = False

    def _log(self, *a):
        if not self.quiet:
            print(*a)

    # ---- hooks implemented by phyloHMRF ----------------------------------------------------------
    def _init(self, X, lengths=None):
        init = 1.0 / self.n_components
        self.startprob_ = np.full(self.n_components, init)
        self.transmat_ = np.full((self.n_components, self.n_components), init)

    def _check(self):
        """startprob_/transmat_ validation (base.py:515-538)."""
        self.startprob_ = np.asarray(self.startprob_)
        if len(self.startprob_) != self.n_components:
            raise ValueError("startprob_ must have length n_components")
        if not np.allclose(self.startprob_.sum(), 1.0):
            raise ValueError("startprob_ must sum to 1.0 (got {0:.4f})".format(self.startprob_.sum()))
        self.transmat_ = np.asarray(self.transmat_)
        if self.transmat_.shape != (self.n_components, self.n_components):
            raise ValueError("transmat_ must have shape (n_components, n_components)")
        if not np.allclose(self.transmat_.sum(axis=1), 1.0):
            raise ValueError("rows of transmat_ must sum to 1.0 (got {0})".format(self.transmat_.sum(axis=1)))

    def _initialize_sufficient_statistics(self):
        return {"nobs": 0, "start": np.zeros(self.n_components),
                "trans": np.zeros((self.n_components, self.n_components))}

    def _accumulate_sufficient_statistics_1(self, stats, stats1):
        """base.py:571-580."""
        stats["post"] += stats1["post"]
        stats["obs"] += stats1["obs"]
        stats["obs*obs.T"] += stats1["obs*obs.T"]
        return stats

    def _do_mstep(self, stats):
        raise NotImplementedError

    def _estep_region(self, region_id):
        """-> (stats dict, cost numerators[4] summed over the region's nodes).  Labels stay on the device."""
        raise NotImplementedError

    def _estep_tiles(self):
        """-> [(stats dict, cost numerators[4], owned nodes)] of the row tiles this rank holds (none by default)"""
        return []

    # ---- the fit loop ----------------------------------------------------------------------------
    def fit_accumulate_test(self, X, len_vec, threshold, annotation, m_iter, lengths=None):
        """Estimate model parameters.  Returns
        (params_vec, params_vec1, params_vecList, iter_id1, iter_id2, cost_vec, t_labels)   (base.py:455)."""
        self._log("Initilization...")
        start = time.time()
        self._init(X, lengths=lengths)
        self._log("use time %s:" % (time.time() - start))
        self._check()
        self._log("model fitting...")
        max_iter = m_iter
        max_iter1 = 50                                   # iterations after the previous minimum (base.py:319)
        pairwise_cost_pre, unary_cost_pre, cost1_pre = 0.001, 0.001, 0.001
        threshold1, threshold2 = threshold, threshold
        cost_vec = []
        min_cost = [0, 1000]                             # base.py:326-327: costs >= 1000 never register
        min_cost1 = [0, 1000]
        params_vec = self.params_vec1.copy()
        params_vec1 = self.params_vec1.copy()
        num_region = len(len_vec)
        n_samples = int(sum(int(len_vec[i][0]) for i in range(num_region)))
        params_vecList = []
        have_t_labels = False
        K, S = self.n_components, self.n_features
        self.timing_ = {"estep": [], "mstep": []}

        self._log("n_iter, m_iter: %d %d" % (self.n_iter, max_iter))
        for it in range(max_iter):
            self._log(it)
            stats = self._initialize_sufficient_statistics()
            start = time.time()
            # E-step of the regions this rank owns; un-normalised cost sums travel with the statistics
            local = np.zeros(K * (1 + S + S * S) + 5)
            by_size = sorted(self.my_regions, key=lambda r: -int(len_vec[r][0]))          # largest block first
            pending = self.runner.start(self._estep_region, by_size)                      # concurrent streams
            tiled = self._estep_tiles()         # row tiles of split blocks: lockstep rounds, on this thread meanwhile
            done = dict(zip(by_size, pending.results()))
            for region_id in self.my_regions:                                             # fixed summation order
                st, costs = done[region_id]
                local[:-5] += pack_stats(st)
                local[-5:-1] += costs
                local[-1] += int(len_vec[region_id][0])
            for st, costs, n_owned in tiled:
                local[:-5] += pack_stats(st)
                local[-5:-1] += costs
                local[-1] += int(n_owned)
            total = self.reducer.allreduce(local)
            self.timing_["estep"].append(time.time() - start)
            self._log("use time %d:" % it)
            self._log(time.time() - start)
            stats = self._accumulate_sufficient_statistics_1(stats, unpack_stats(total[:-5], K, S))
            # sum_r cost_r * n_r / N  (base.py:388-391)  ==  (sum of un-normalised sums) / N
            pairwise_cost1, pairwise_cost, unary_cost, cost1 = (total[-5:-1] / n_samples).tolist()

            t_difference1 = abs((pairwise_cost - pairwise_cost_pre) * 1.0 / pairwise_cost_pre)
            t_difference2 = abs((unary_cost - unary_cost_pre) * 1.0 / unary_cost_pre)
            t_difference3 = abs((cost1 - cost1_pre) * 1.0 / cost1_pre)
            self._log(pairwise_cost_pre, pairwise_cost, unary_cost_pre, unary_cost, cost1_pre, cost1)
            self._log(t_difference1, t_difference2, t_difference3)
            pairwise_cost_pre, unary_cost_pre, cost1_pre = pairwise_cost, unary_cost, cost1
            cost_vec.append([it, pairwise_cost, unary_cost, cost1])       # base.py:410
            params_vecList.append(self.params_vec1.copy())

            if cost1 < min_cost[1]:                                        # base.py:416-420
                min_cost = [it, cost1]
                params_vec = self.params_vec1.copy()
                self._snapshot_labels(SLOT_LOCAL)
                self._log("another temp min")
            if cost1 < min_cost1[1] and it >= 3:                           # base.py:422-426
                min_cost1 = [it, cost1]
                params_vec1 = self.params_vec1.copy()
                self._snapshot_labels(SLOT_BEST3)
                have_t_labels = True
                self._log("another temp min from iteration 3")
            if ((t_difference1 < threshold1 and t_difference2 < threshold2) or (t_difference3 < threshold1)) and (it > 5):
                break                                                      # base.py:428-429
            if it > max_iter:
                break
            if it - min_cost1[0] > max_iter1:                              # base.py:434-435
                break
            self._log("Maximization...")
            start = time.time()
            self._do_mstep(stats)
            self.timing_["mstep"].append(time.time() - start)
            self._log("maximization use time %d %s" % (it, time.time() - start))

        self.params_vec1 = params_vec1.copy()                              # base.py:444
        self._ou_param_varied_constraint(params_vec)                       # base.py:445
        cost_vec = np.asarray(cost_vec)
        self._log(min_cost)
        self._log(cost_vec)
        params_vecList = np.asarray(params_vecList)
        t_labels = self._gather_labels(SLOT_BEST3) if have_t_labels else np.zeros(n_samples)   # base.py:342
        return params_vec, params_vec1, params_vecList, min_cost[0], min_cost1[0], cost_vec, t_labels
