"""EM driver with the reference's fit surface (`_BaseGraph.fit_accumulate_test`, base.py:301-455).

What is kept: the method name, argument order, return tuple, the per-iteration bookkeeping (cost weighting by
n_r/N, cost_vec rows, min_cost / min_cost1 tracking, labels_local warm start, t_labels from iteration 3 on,
the three stopping rules, no M-step after the last E-step) and the per-iteration prints.
New (optional, SURVEY.md section 5 "checkpoint/resume": the reference keeps everything in RAM only, base.py:412): a
checkpoint per EM iteration (`checkpoint_path`, `checkpoint_every`) and `resume_from` -- the loop restarts at the saved
iteration with the same bookkeeping state (base.py:402-435), labels snapshots and random generator.
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
        # optional .npz checkpoint of the fit (written by rank 0 after the M-step of every `checkpoint_every`-th iteration,
        # atomically: temporary file + rename) and the file a fit resumes from (read by every rank)
        self.checkpoint_path = None
        self.checkpoint_every = 1
        self.checkpoint_labels = True
        self.resume_from = None
        self.em_iteration_ = -1

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

    def _prepare_next_estep(self):
        """hook: queue device work of the next E-step that does not depend on the M-step's result; -> None or an object with
        .results() that waits for the queueing (not for the device)"""
        return None

    def _estep_region(self, region_id):
        """-> (stats dict, cost numerators[4] summed over the region's nodes).  Labels stay on the device."""
        raise NotImplementedError

    def _estep_tiles(self):
        """-> [(stats dict, cost numerators[4], owned nodes)] of the row tiles this rank holds (none by default)"""
        return []

    def _estep_lockstep(self, regions):
        """hook: the E-step of all `regions` driven from the calling thread in lockstep rounds -> {region: (stats, costs)},
        or None when the model runs its regions on the runner's threads (the default)"""
        return None

    # ---- checkpoint / resume -----------------------------------------------------------------------
    CHECKPOINT_FORMAT = 1

    def _write_checkpoint(self, it_next, loop):
        """State of the fit after the M-step of iteration it_next - 1: everything the loop of fit_accumulate_test carries
        (base.py:402-435), the model parameters, the random generator, and -- unless checkpoint_labels is off -- the three
        labellings a resumed fit needs (labels_local, t_labels, the previous E-step's result).  Collective when the labels
        are included (they are gathered from the ranks); rank 0 writes."""
        import json
        import os
        d = dict(format=self.CHECKPOINT_FORMAT, it_next=it_next, n_components=self.n_components, n_features=self.n_features,
                 params_vec1_model=self.params_vec1, means_=self.means_, covars_=self._covars_,
                 init_ou_params=self.init_ou_params, cost_vec=np.asarray(loop["cost_vec"], dtype=np.float64).reshape(-1, 4),
                 params_vecList=np.asarray(loop["params_vecList"]), min_cost=np.asarray(loop["min_cost"], dtype=np.float64),
                 min_cost1=np.asarray(loop["min_cost1"], dtype=np.float64), params_vec_best=loop["params_vec"],
                 params_vec1_best=loop["params_vec1"], pre=np.asarray(loop["pre"], dtype=np.float64),
                 have_t_labels=int(loop["have_t_labels"]), rng_state=json.dumps(self.rng.bit_generator.state),
                 len_vec=np.asarray(self.len_vec), model_key=json.dumps(self._checkpoint_key(), sort_keys=True))
        if self.checkpoint_labels:
            self._snapshot_labels(SLOT_CURRENT)
            d["labels_current"] = self._gather_labels(SLOT_CURRENT).astype(np.uint8)
            d["labels_local"] = self._gather_labels(SLOT_LOCAL).astype(np.uint8)
            if loop["have_t_labels"]:
                d["labels_best3"] = self._gather_labels(SLOT_BEST3).astype(np.uint8)
        if getattr(self, "rank", 0) == 0:
            tmp = self.checkpoint_path + ".tmp.npz"
            np.savez(tmp, **d)
            os.replace(tmp, self.checkpoint_path)

    def _checkpoint_key(self):
        """what the saved cost bookkeeping (cost_vec, min_cost, the stopping rules) is only meaningful under: the energy's
        coefficients, the cost variant, the start policy and solver options, and a cheap fingerprint of the observations
        (a checkpoint of the same block sizes but other data or another beta must not be resumed silently)"""
        key = {}
        for name in ("beta", "beta1", "estimate_type", "num_neighbor", "warm_start", "lambda_0", "min_covar"):
            v = getattr(self, name, None)
            key[name] = v if isinstance(v, (str, type(None))) else float(v)
        key["solver_opts"] = {k: (v if isinstance(v, (str, bool)) else float(v)) for k, v in sorted(getattr(self, "solver_opts", {}).items())}
        X = getattr(self, "observation", None)
        if X is not None:
            X = np.asarray(X)
            idx = np.linspace(0, X.shape[0] - 1, num=min(X.shape[0], 4096)).astype(np.int64)
            key["observation"] = [int(X.shape[0]), int(X.shape[1]), repr(float(np.asarray(X[idx], dtype=np.float64).sum()))]
        return key

    def _read_checkpoint(self, path, loop):
        """-> the iteration to continue with; fills `loop` and the model from the file, puts the labellings back on the device"""
        import json
        z = np.load(path, allow_pickle=False)
        if int(z["format"]) != self.CHECKPOINT_FORMAT:
            raise ValueError("checkpoint %s has format %d, this build reads %d" % (path, int(z["format"]), self.CHECKPOINT_FORMAT))
        if int(z["n_components"]) != self.n_components or int(z["n_features"]) != self.n_features or \
                not np.array_equal(np.asarray(z["len_vec"]), np.asarray(self.len_vec)):
            raise ValueError("checkpoint %s belongs to another model or data set (states, species or len_vec differ)" % path)
        if "model_key" in z.files:           # (files of round 5 carry none: nothing to compare)
            saved, now = json.loads(str(z["model_key"])), json.loads(json.dumps(self._checkpoint_key(), sort_keys=True))
            diff = sorted(k for k in set(saved) | set(now) if saved.get(k) != now.get(k))
            if diff:
                raise ValueError("checkpoint %s was written under another model or data (%s differ: saved %s, now %s): its cost "
                                 "bookkeeping and stopping state do not carry over -- start a new fit"
                                 % (path, ", ".join(diff), {k: saved.get(k) for k in diff}, {k: now.get(k) for k in diff}))
        if "labels_local" not in z.files:
            raise ValueError("checkpoint %s was written without the labellings (checkpoint_labels=False): parameters can be "
                             "read from it, a fit cannot resume from it" % path)
        self.params_vec1 = z["params_vec1_model"].copy()
        self.means_, self._covars_ = z["means_"].copy(), z["covars_"].copy()
        self.init_ou_params = z["init_ou_params"].copy()
        loop["cost_vec"] = [list(r) for r in z["cost_vec"].tolist()]
        for r in loop["cost_vec"]:
            r[0] = int(r[0])
        loop["params_vecList"] = [p.copy() for p in z["params_vecList"]]
        loop["min_cost"] = [int(z["min_cost"][0]), float(z["min_cost"][1])]
        loop["min_cost1"] = [int(z["min_cost1"][0]), float(z["min_cost1"][1])]
        loop["params_vec"], loop["params_vec1"] = z["params_vec_best"].copy(), z["params_vec1_best"].copy()
        loop["pre"] = z["pre"].tolist()
        loop["have_t_labels"] = bool(int(z["have_t_labels"]))
        self.rng.bit_generator.state = json.loads(str(z["rng_state"]))
        if loop["have_t_labels"]:
            self._upload_labels_slot(np.int64(z["labels_best3"]), SLOT_BEST3)
        self._upload_labels_slot(np.int64(z["labels_local"]), SLOT_LOCAL)
        self._upload_labels_slot(np.int64(z["labels_current"]), None)        # last: the blocks' current labelling
        return int(z["it_next"])

    # ---- the fit loop ----------------------------------------------------------------------------
    def fit_accumulate_test(self, X, len_vec, threshold, annotation, m_iter, lengths=None):
        """Estimate model parameters.  Returns
        (params_vec, params_vec1, params_vecList, iter_id1, iter_id2, cost_vec, t_labels)   (base.py:455)."""
        self._log("Initilization...")
        start = time.time()
        resumed = {}
        it0 = 0
        if self.resume_from:
            _BaseGraph._init(self, X, lengths=lengths)    # (start / transition vectors only: the rest comes from the file)
            it0 = self._read_checkpoint(self.resume_from, resumed)
        else:
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
        self.timing_ = {"estep": [], "mstep": [], "iteration": []}       # seconds; "iteration": the whole loop body, bookkeeping included

        if self.resume_from:
            loop = resumed
            cost_vec, params_vecList = loop["cost_vec"], loop["params_vecList"]
            min_cost, min_cost1 = loop["min_cost"], loop["min_cost1"]
            params_vec, params_vec1 = loop["params_vec"], loop["params_vec1"]
            pairwise_cost_pre, unary_cost_pre, cost1_pre = loop["pre"]
            have_t_labels = loop["have_t_labels"]
            self._log("resumed from %s at iteration %d" % (self.resume_from, it0))
        self._log("n_iter, m_iter: %d %d" % (self.n_iter, max_iter))
        for it in range(it0, max_iter):
            self._log(it)
            self.em_iteration_ = it
            stats = self._initialize_sufficient_statistics()
            start = time.time()
            t_iteration = start
            # E-step of the regions this rank owns; un-normalised cost sums travel with the statistics
            local = np.zeros(K * (1 + S + S * S) + 5)
            by_size = sorted(self.my_regions, key=lambda r: -int(len_vec[r][0]))          # largest block first
            if getattr(self, "lockstep", False):
                # block_threads=0: every whole block from ONE thread (Block.solve_group) -- this one, unless the rank also holds
                # row tiles: their lockstep rounds run here then, and the whole blocks on one helper thread meanwhile
                if by_size and getattr(self, "tile_groups", None):
                    pending = self.runner.start_one(lambda: self._estep_lockstep(by_size))
                    tiled = self._estep_tiles()
                    done = pending.results()[0]
                else:
                    done = self._estep_lockstep(by_size)
                    tiled = self._estep_tiles()
            else:
                pending = self.runner.start(self._estep_region, by_size)                  # concurrent streams
                tiled = self._estep_tiles()     # row tiles of split blocks: lockstep rounds, on this thread meanwhile
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
                self.timing_["iteration"].append(time.time() - t_iteration)
                break                                                      # base.py:428-429
            if it > max_iter:
                self.timing_["iteration"].append(time.time() - t_iteration)
                break
            if it - min_cost1[0] > max_iter1:                              # base.py:434-435
                self.timing_["iteration"].append(time.time() - t_iteration)
                break
            self._log("Maximization...")
            start = time.time()
            ahead = self._prepare_next_estep()       # (device work that needs the labels only, queued behind the host's M-step)
            self._do_mstep(stats)
            if ahead is not None:
                ahead.results()
            self.timing_["mstep"].append(time.time() - start)
            self.timing_["iteration"].append(time.time() - t_iteration)
            self._log("maximization use time %d %s" % (it, time.time() - start))
            if self.checkpoint_path and (it + 1) % max(int(self.checkpoint_every), 1) == 0:
                self._write_checkpoint(it + 1, dict(cost_vec=cost_vec, params_vecList=params_vecList, min_cost=min_cost,
                                                    min_cost1=min_cost1, params_vec=params_vec, params_vec1=params_vec1,
                                                    pre=[pairwise_cost_pre, unary_cost_pre, cost1_pre],
                                                    have_t_labels=have_t_labels))

        self.params_vec1 = params_vec1.copy()                              # base.py:444
        self._ou_param_varied_constraint(params_vec)                       # base.py:445
        cost_vec = np.asarray(cost_vec)
        self._log(min_cost)
        self._log(cost_vec)
        params_vecList = np.asarray(params_vecList)
        t_labels = self._gather_labels(SLOT_BEST3) if have_t_labels else np.zeros(n_samples)   # base.py:342
        return params_vec, params_vec1, params_vecList, min_cost[0], min_cost1[0], cost_vec, t_labels
