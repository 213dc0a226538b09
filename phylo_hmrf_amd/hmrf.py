"""`phyloHMRF` with the reference's constructor / fit / predict surface (phylo_hmrf.py:51-1528), E-step on MI355X.

Per syntenic block one `Block` holds X, the neighbour graph, logprob and labels in HBM for the whole fit
(the reference recomputes / re-pickles them per iteration, base.py:357-372).  Kept names and return shapes:
  _compute_log_likelihood(X)                                  phylo_hmrf.py:266-268
  predict(X, region_id) -> (state, logprob)                   :470-484
  _estimate_state_graphcuts_gco(X, init, edges, weights)      :486-507   (GPU label solver instead of pygco/gco)
  _compute_posteriors_graph(X, label, logprob, region_id)     :334-355
  _predict_posteriors(X, len_vec, region_id, m_queue=None)    :297-322
  _ou_param_varied_constraint(params_vec), _do_mstep(stats), _init(X), fit_accumulate_test(...)
"""
from __future__ import print_function

import numpy as np

from . import mstep as _mstep
from .base import SLOT_BEST3, SLOT_LOCAL, _BaseGraph
from ._lib import PhmrfError
from .block import Block
from .dist import Reducer, env_rank_world, lpt_assign
from .tree import PhyloTree

COVARIANCE_TYPES = frozenset(("linear", "spherical", "diag", "full", "tied"))


class phyloHMRF(_BaseGraph):

    def __init__(self, n_samples, n_features, edge_list, branch_list, cons_param, beta, beta1,
                 initial_mode, initial_weight, initial_weight1, initial_magnitude, observation,
                 edge_list_1, len_vec, type_id=0, max_iter=10, n_components=1, run_id=0, estimate_type=0,
                 covariance_type="full", min_covar=1e-3, startprob_prior=1.0, transmat_prior=1.0,
                 means_prior=0, means_weight=0, covars_prior=1e-2, covars_weight=1, algorithm="viterbi",
                 random_state=None, n_iter=10, tol=1e-2, verbose=False, params="stmc", init_params="stmc",
                 learning_rate=0.001, num_neighbor=8, block_factory=None, reducer=None, world=None, rank=None,
                 solver_opts=None, mstep_workers=None, quiet=False, device_graph=False, block_threads=14,
                 init_method="minibatch", split_above=1.0, tile_parts=None, warm_start="best", checkpoint_path=None,
                 checkpoint_every=1, checkpoint_labels=True, resume_from=None):
        _BaseGraph.__init__(self, n_components=n_components, run_id=run_id, estimate_type=estimate_type,
                            startprob_prior=startprob_prior, transmat_prior=transmat_prior, algorithm=algorithm,
                            random_state=random_state, n_iter=n_iter, tol=tol, params=params, verbose=verbose,
                            init_params=init_params)
        if covariance_type not in COVARIANCE_TYPES:
            raise ValueError("covariance_type must be one of {0}".format(COVARIANCE_TYPES))
        if n_components < 1 or n_components > 64:
            # one byte per label on the device and one bit per label in the 64-bit label masks (phmrf_block_create)
            raise ValueError("phyloHMRF on the GPU supports 1..64 states (n_components = %d)" % n_components)
        if n_features > 8:
            # the posterior / statistics kernel holds a node's [1 | x | x x^T] features in LDS for S <= 8 (kernels.hip);
            # the emission kernel alone goes to S = 16 (phmrf_emission_dev)
            raise ValueError("phyloHMRF on the GPU supports at most 8 species (n_features = %d)" % n_features)
        self.quiet = quiet
        self.checkpoint_path, self.checkpoint_every = checkpoint_path, checkpoint_every
        self.checkpoint_labels, self.resume_from = checkpoint_labels, resume_from
        self.covariance_type = covariance_type
        self.min_covar = min_covar
        self.max_iter = max_iter
        self.beta = beta              # Potts coefficient (phylo_hmrf.py:84)
        self.beta1 = beta1            # edge-weight decay w = exp(-beta1*d) (:85, :585)
        self.type_id = type_id
        self.observation = observation
        self.n_samples = n_samples
        self.n_features = n_features
        self.learning_rate = learning_rate
        self.edge_list_vec = edge_list_1
        self.len_vec = [list(map(int, lv)) for lv in np.asarray(len_vec).tolist()]
        self.num_neighbor = num_neighbor
        self.edge_potential = self._pairwise_potential()
        # M-step workers are forked ONCE, before this process touches the GPU (a fork after HIP / RCCL / the block threads
        # are up is unsafe); later requests for fewer workers reuse the same pool, close() ends it
        self.mstep_workers = mstep_workers
        env_rank, env_world, _ = env_rank_world()
        self.world = env_world if world is None else world
        self.rank = env_rank if rank is None else rank
        if block_factory is None and not _mstep.native_available():
            # (only when the native M-step -- all states in one library call on host threads -- is unavailable: the Python
            #  fall-back fits the states in worker processes, which must be forked before this process touches the GPU)
            import os as _os
            want = min(n_components, _os.cpu_count() or 1) if mstep_workers is None else int(mstep_workers)
            _mstep._pool(want)
        self.reducer = reducer if reducer is not None else Reducer(self.world)
        if self.world > 1:
            # every rank must draw the same initial parameters, k-means seed and restarts: rank 0's seed wins
            # (random_state None: rank 0 draws one)
            seed0 = float(np.random.SeedSequence().entropy % (2 ** 31)) if random_state is None else float(random_state)
            random_state = int(self.reducer.broadcast(np.array([seed0]), src=0)[0])
            self.random_state = random_state
        self.rng = np.random.default_rng(random_state)
        # the label solver stops when a whole round improves the energy by less than 1e-5 of it (energy_tol_ppb = 0:
        # until a verification round with every move type finds nothing -- both are below the reference's gco result).
        # 1e-5 is the level at which two solves of the same inputs differ (the phase of their cuts, the order of their
        # atomics: 1.4e-5 per E-step); until round 6 the default was 1e-6, which costs 8 % more E-step time for 5e-6 +- 2e-6
        # of energy on the same inputs, where either stops 1.7e-4 above the exact fixed point
        self.solver_opts = dict(max_rounds=64, use_chains=True, use_components=True, use_strips=True, use_expansion=True,
                                energy_tol_ppb=10000)
        if solver_opts:
            self.solver_opts.update(solver_opts)
        if init_method not in ("minibatch", "sklearn", "device"):
            raise ValueError("init_method must be 'minibatch' (the reference's MiniBatchKMeans for the centres, every pass "
                             "over all nodes on the device), 'sklearn' (the reference's initialisation verbatim, on the "
                             "host) or 'device' (Lloyd iterations on the device)")
        self.init_method = init_method
        # the start of an E-step's labelling: "local" = labels_local, the labels of the iteration with the lowest cost so far,
        # as the reference does (phylo_hmrf.py:479, base.py:416-420); "best" = that or the previous E-step's labels, whichever
        # has the lower energy under the new parameters (Block.warm_start) -- labels_local can be dozens of iterations old
        if warm_start not in ("best", "local"):
            raise ValueError("warm_start must be 'best' or 'local'")
        self.warm_start = warm_start

        # species tree tables (phylo_hmrf.py:103-143)
        self.tree = PhyloTree(edge_list)
        self.node_num = self.tree.node_num
        self.branch_params = branch_list
        self.branch_dim = self.tree.branch_dim
        self.n_params = self.tree.n_params
        self.leaf_vec = self.tree.leaf_vec
        if self.tree.n_features != n_features:
            raise ValueError("tree has %d leaves but n_features is %d" % (self.tree.n_features, n_features))
        self.params_vec1 = self.rng.random((n_components, self.n_params))     # :109
        self.init_ou_params = self.params_vec1.copy()
        self.lambda_0 = cons_param                                             # :145
        self.initial_mode, self.initial_w1, self.initial_w1a, self.initial_w2 = (
            initial_mode, initial_weight, initial_weight1, initial_magnitude)
        self.stats = dict()

        # block ownership: whole syntenic blocks are dealt to the ranks longest first; a block that holds more than
        # split_above x (all nodes / world) is cut into row tiles first (tiles.py), which are dealt like blocks
        from . import tiles as _tiles
        geo = []
        for lv in self.len_vec:
            n, H, W = lv[0], lv[3], lv[4]
            diag = bool(lv[8]) if len(lv) > 8 else (H == W and n == H * (H + 1) // 2)
            geo.append((H, W, diag, n == (H * (H + 1) // 2 if diag else H * W) and num_neighbor == 8))
        can_split = block_factory is None
        force = {int(r): int(p) for r, p in (tile_parts or {}).items()} if can_split else {}
        units = []
        share = sum(lv[0] for lv in self.len_vec) / float(max(self.world, 1))
        for r, (H, W, diag, grid_ok) in enumerate(geo):
            parts = 1
            if can_split and grid_ok:
                if r in force:
                    parts = force[r]
                elif self.world > 1 and self.len_vec[r][0] > split_above * share:
                    parts = int(np.ceil(self.len_vec[r][0] / (split_above * share)))
                # a block with a host edge list is only cut if that list IS the grid stencil -- every rank asks the same
                # question of the same list and gets the same answer; a block that fails stays whole and takes the
                # general-graph fallback below (a RuntimeWarning) instead of aborting in make_group on its holders only
                if parts > 1 and not device_graph and not _tiles.edges_fit_grid(
                        np.asarray(edge_list_1[r])[:, 0:2], H, W, diag, num_neighbor):
                    parts = 1
            rows = _tiles.split_rows(H, W, diag, parts) if parts > 1 else [(0, H)]
            if len(rows) == 1:
                units.append(dict(block=r, tile=0, ntiles=1, r0=0, r1=H, nodes=self.len_vec[r][0]))
            else:
                for t, (r0, r1) in enumerate(rows):
                    units.append(dict(block=r, tile=t, ntiles=len(rows), r0=r0, r1=r1, nodes=_tiles.rows_nodes(r0, r1, W, diag)))
        unit_owner = _tiles.assign(units, self.world)
        self.units, self.unit_owner = units, unit_owner
        self.owner = [unit_owner[i] for i, u in enumerate(units) if u["tile"] == 0]      # (of a split block: its first tile's)
        self.my_regions = [u["block"] for i, u in enumerate(units) if u["ntiles"] == 1 and unit_owner[i] == self.rank]
        self.split_regions = sorted(set(u["block"] for u in units if u["ntiles"] > 1))

        # one process per GPU: this rank's device is LOCAL_RANK (blocks, their streams and the worker threads use it)
        device = None
        if block_factory is None:
            from . import _lib
            _, _, device = env_rank_world()
            _lib.require_gpu()
            _lib.check(_lib.load().phmrf_set_device(device))
            if self.world > 1:
                import torch
                torch.cuda.set_device(device)
        # independent blocks run concurrently, each on its own stream (the reference: one process per block, base.py:357)
        from .concurrent import BlockRunner
        # block_threads=0: no thread pool -- this thread drives every whole block, round by round as the rounds end (phmrf_mrf_solve_group)
        self.lockstep = int(block_threads) == 0
        self.runner = BlockRunner(min(max(int(block_threads), 1), max(1, len(self.my_regions))), device)

        # device-resident blocks (the reference's _edge_weight_undirected_vec, :567-598, happens here once)
        factory = block_factory or Block
        self.blocks = {}
        self.general_graph_regions = []      # regions whose edge list is not the grid stencil: no chain / strip moves
        X = np.asarray(observation)
        for r in self.my_regions:
            lv = self.len_vec[r]
            n, s1, s2, H, W = lv[0], lv[1], lv[2], lv[3], lv[4]
            diag = geo[r][2]
            b = factory(n, n_features, n_components)
            b.set_observations(X[s1:s2])
            grid_ok = (n == (H * (H + 1) // 2 if diag else H * W)) and num_neighbor in (4, 8)
            if device_graph and grid_ok:
                b.build_grid_graph(H, W, diag, num_neighbor, beta1)
            else:
                e = np.asarray(edge_list_1[r])
                w = np.exp(-beta1 * e[:, 2])                                    # :585
                b.set_graph(np.int64(e[:, 0:2]), w)                             # :589
                if grid_ok:
                    try:
                        b.set_grid(H, W, diag, num_neighbor)
                    except PhmrfError as err:
                        if err.status != 1:              # PHMRF_ERR_INVALID = "edges do not fit the grid"; anything
                            raise                        # else (HIP, allocation) is a real failure
                        grid_ok = False
                if not grid_ok:
                    import warnings
                    warnings.warn("region %d: the edge list / len_vec is not a contact-map grid block (utility.py:1871-"
                                  "2053); it is labelled with general-graph moves only (ICM, component and path moves, alpha-"
                                  "expansion by minimum cut): host-synchronous and far slower per node than the grid "
                                  "path; energy parity with gco's swap is tested on k-NN graphs (tests/test_gpu_estep.py)"
                                  % r, RuntimeWarning)
                    self.general_graph_regions.append(r)
            self.blocks[r] = b

        # row tiles of the split blocks: this rank's tiles, and per split block the group of ranks that hold its tiles
        # (every rank creates every group, in block order: torch.distributed.new_group is collective)
        self.tile_groups = {}
        for r in self.split_regions:
            mine = [(i, u) for i, u in enumerate(units) if u["block"] == r]
            rows = [(u["r0"], u["r1"]) for _, u in mine]
            owners = [unit_owner[i] for i, _ in mine]
            comm = None
            if self.world > 1:
                dev_t = None
                import torch.distributed as dist
                if dist.get_backend() == "nccl":
                    import torch
                    dev_t = torch.device("cuda", torch.cuda.current_device())
                comm = _tiles.GroupComm(owners, dev_t)
            if self.rank not in owners:
                continue
            lv = self.len_vec[r]
            s1, H, W, diag = lv[1], geo[r][0], geo[r][1], geo[r][2]
            e_all = None if device_graph else np.asarray(edge_list_1[r])

            def load(tl, s1=s1, e_all=e_all):
                tl.b.set_observations(X[s1 + tl.node0:s1 + tl.node0 + tl.n])

            grp = _tiles.make_group(r, (H, W, diag), rows, owners, self.rank, n_features, n_components, factory, load, comm,
                                    num_neighbor, beta1, edges=e_all)
            self.tile_groups[r] = grp
        self.conductor = _tiles.Conductor(list(self.tile_groups.values()))

    def _local_units(self):
        """(block, slice of the samples its owned nodes are, slice of its own node ids that are owned) of every whole block
        and every tile this rank holds"""
        out = []
        for r in self.my_regions:
            s1, s2 = self.len_vec[r][1], self.len_vec[r][2]
            out.append((self.blocks[r], slice(s1, s2), slice(0, s2 - s1), slice(s1, s2)))
        for r, grp in self.tile_groups.items():
            s1 = self.len_vec[r][1]
            for t in sorted(grp.local):
                tl = grp.local[t]
                out.append((tl.b, slice(s1 + tl.node0 + tl.own_lo, s1 + tl.node0 + tl.own_hi), tl.owned_local_slice(),
                            slice(s1 + tl.node0, s1 + tl.node0 + tl.n)))
        return out

    # ---- properties the reference exposes --------------------------------------------------------
    def _get_covars(self):
        return self._covars_

    def _set_covars(self, covars):
        self._covars_ = np.asarray(covars).copy()

    covars_ = property(_get_covars, _set_covars)

    def _pairwise_potential(self):
        """beta * (1 - delta)  (phylo_hmrf.py:524-536)."""
        V = np.full((self.n_components, self.n_components), float(self.beta))
        np.fill_diagonal(V, 0.0)
        self.edge_potential = V
        return V

    def _initialize_sufficient_statistics(self):
        """phylo_hmrf.py:691-698."""
        stats = super(phyloHMRF, self)._initialize_sufficient_statistics()
        K, S = self.n_components, self.n_features
        stats["post"] = np.zeros(K)
        stats["obs"] = np.zeros((K, S))
        stats["obs**2"] = np.zeros((K, S))
        stats["obs*obs.T"] = np.zeros((K, S, S))
        return stats

    def _check(self):
        super(phyloHMRF, self)._check()
        self.means_ = np.asarray(self.means_)
        self.n_features = self.means_.shape[1]

    # ---- initialisation (phylo_hmrf.py:205-264) --------------------------------------------------
    def _init(self, X, lengths=None):
        super(phyloHMRF, self)._init(X, lengths=lengths)
        X = np.asarray(X)
        n_samples, n_features = X.shape
        seed = None if self.random_state is None else int(self.random_state)
        my_blocks = [u[0] for u in self._local_units()]
        red = self.reducer if self.world > 1 else None
        if self.init_method in ("minibatch", "device"):
            from .kmeans import device_kmeans, device_moments, minibatch_centers
            if self.init_method == "minibatch":
                # the reference's estimator and settings for the centres (:234-236); every rank computes the same ones
                centers = minibatch_centers(X, self.n_components, seed)
            else:
                # Lloyd iterations on the GPU (kmeans.py); every rank draws the same host sample for seeding
                srng = np.random.default_rng(seed)
                rows = srng.choice(n_samples, size=min(n_samples, 50000), replace=False)
                centers, _ = device_kmeans(my_blocks, X[np.sort(rows)], self.n_components, srng, reducer=red)
            # what the reference does next with passes over all rows -- labels_ (:239), the per-cluster OU fit on a cluster's
            # mean and X^T X / n (:246 -> :1246-1325), the global covariance (:258) -- from ONE device pass over X
            counts, sums, outer, _ = device_moments(my_blocks, centers, reducer=red, write_labels=True)
            self.means_ = centers
            self._snapshot_labels(SLOT_LOCAL)
            init_label = np.int64(self._gather_labels(SLOT_LOCAL))
            self._log("initialize parameters...")
            self.init_ou_params = _mstep.init_ou_params_moments(self.tree, counts, sums, outer, self.means_, self.params_vec1,
                                                                self.initial_w2, self.rng, workers=self.mstep_workers)
            n_tot = float(counts.sum())
            mean_all = sums.sum(axis=0) / n_tot
            cv = (outer.sum(axis=0) - n_tot * np.outer(mean_all, mean_all)) / (n_tot - 1.0) + self.min_covar * np.eye(n_features)
        else:
            from sklearn import cluster
            kmeans = cluster.MiniBatchKMeans(n_clusters=self.n_components, random_state=seed, batch_size=2000,
                                             max_iter=1000, n_init=10)          # :234-236
            kmeans.fit(X)
            self.means_ = kmeans.cluster_centers_
            init_label = kmeans.labels_
            self._log("initialize parameters...")
            self.init_ou_params = _mstep.init_ou_params(self.tree, X, init_label, self.means_, self.params_vec1,
                                                        self.initial_w2, self.rng, workers=self.mstep_workers)   # :246
            cv = np.cov(X.T) + self.min_covar * np.eye(n_features)              # :258
        self.params_vec1 = self.init_ou_params.copy()
        self.init_label = np.int64(init_label)
        self.labels = self.init_label.copy()
        self.labels_local = self.init_label.copy()
        self._upload_labels(self.init_label)
        self._covars_ = np.tile(np.atleast_2d(cv), (self.n_components, 1, 1))   # :261-262
        self._log("return from initializing parameters...")

    def _upload_labels(self, labels):
        for b, _, _, stored in self._local_units():          # (a tile takes its halo rows too)
            b.set_labels(np.asarray(labels[stored]))
            b.save_labels(SLOT_LOCAL)

    def _upload_labels_slot(self, labels, slot):
        """a labelling of all samples -> the blocks' snapshot `slot` (None: their current labels); a tile takes its halo rows"""
        for b, _, _, stored in self._local_units():
            b.set_labels(np.asarray(labels[stored]))
            if slot is not None:
                b.save_labels(slot)

    def _snapshot_labels(self, slot):
        for b, _, _, _ in self._local_units():
            b.save_labels(slot)

    def _gather_labels(self, slot):
        """the labelling of all samples from the blocks' snapshot `slot`, as float64 like the reference's `labels`
        (base.py:381, :394).  One byte per label until the last line (K <= 64): with several ranks the exchange is an
        all-reduce of n_samples BYTES -- every position is written by exactly one rank, the others add zeros -- where it was one
        of n_samples doubles (710 MB at the whole-genome workload's 88.8 M nodes)."""
        out = np.zeros(self.n_samples, dtype=np.uint8)
        for b, own, own_local, _ in self._local_units():
            out[own] = b.get_saved_labels(slot)[own_local]
        if self.world > 1:
            out = self.reducer.allreduce_bytes(out)
        return out.astype(np.float64)

    # ---- b1 --------------------------------------------------------------------------------------
    def _compute_log_likelihood(self, X):
        X = np.asarray(X, dtype=np.float64)
        b = Block(X.shape[0], X.shape[1], self.n_components)
        try:
            b.set_observations(X)
            b.emission(self.means_, self._covars_)
            return b.get_logprob()
        finally:
            b.close()

    # ---- b2 --------------------------------------------------------------------------------------
    def predict(self, X, region_id):
        """labels and emission log-likelihoods of one region, warm-started from labels_local (:470-484)."""
        if region_id in self.split_regions:
            return self._predict_tiled(region_id)
        b = self._whole_block(region_id, "predict")
        b.restore_labels(SLOT_LOCAL)
        b.emission(self.means_, self._covars_)
        b.solve_fast(self.beta, **self.solver_opts)
        return b.get_labels(), b.get_logprob()

    def _predict_tiled(self, region_id):
        """predict() of a region that is cut into row tiles: a collective of THE RANKS THAT HOLD ITS TILES (the tile group's
        communicator -- a rank that holds none of them raises KeyError naming the holders and takes part in nothing, so the
        holders may call it on their own); every holder returns the whole region's labels and log-likelihoods, the other
        holders' rows arrive by an all-reduce over the group"""
        from .tiles import Conductor
        n = self.len_vec[region_id][0]
        grp = self.tile_groups.get(region_id)
        if grp is None:                          # (like _whole_block: a region this rank does not hold is an error, not zeros)
            holders = sorted(set(int(self.unit_owner[i]) for i, u in enumerate(self.units) if u["block"] == region_id))
            raise KeyError("region %d is cut into row tiles held by ranks %s, none of them by this rank (%d of %d)"
                           % (region_id, holders, self.rank, self.world))
        labels, logprob = np.zeros(n, dtype=np.int64), np.zeros((n, self.n_components))

        def prepare(tl):
            tl.b.restore_labels(SLOT_LOCAL)
            tl.b.emission(self.means_, self._covars_)

        Conductor([grp]).solve(self.beta, self.solver_opts, prepare=prepare)
        for tl in grp.local.values():
            labels[tl.owned_global_slice()] = tl.b.get_labels()[tl.owned_local_slice()]
            logprob[tl.owned_global_slice()] = tl.b.get_logprob()[tl.owned_local_slice()]
        if grp.comm is not None:                          # (each row is written by exactly one rank: the sum is a gather)
            labels = grp.comm.allreduce_i64(labels)
            logprob = grp.comm.allreduce_i64(logprob.ravel().view(np.int64)).view(np.float64).reshape(n, self.n_components)
        return labels.astype(np.int32), logprob

    def _whole_block(self, region_id, what):
        """the resident block of a region this rank holds whole -- or the reason why there is none"""
        if region_id in self.split_regions:
            raise NotImplementedError("region %d is cut into row tiles held by several ranks: %s comes out of the fit's E-step "
                                      "(fit_accumulate_test); construct the model with split_above=float('inf') to keep "
                                      "every block whole" % (region_id, what))
        if region_id not in self.blocks:
            raise KeyError("region %d is held by rank %d, not by this rank (%d of %d)"
                           % (region_id, self.owner[region_id], self.rank, self.world))
        return self.blocks[region_id]

    def _estimate_state_graphcuts_gco(self, X, init_labels1, edge_idList_undirected, edge_weightList_undirected):
        """Drop-in for the pygco call (:486-507) on an arbitrary graph: GPU label solver, general-graph moves."""
        from .pygco_compat import cut_general_graph
        logprob = self._compute_log_likelihood(X)
        labels = cut_general_graph(edge_idList_undirected, edge_weightList_undirected, -logprob, self.edge_potential,
                                   n_iter=5000, algorithm="swap", init_labels=init_labels1)
        return labels, logprob

    # ---- b3 --------------------------------------------------------------------------------------
    def _compute_posteriors_graph(self, X, label, logprob, region_id):
        b = self._whole_block(region_id, "its posteriors")
        b.set_logprob(logprob)
        b.set_labels(label)
        _, costs, post = b.posterior_stats(self.beta, self.estimate_type, want_posteriors=True)
        n = float(len(label))
        return post, costs[0] / n, costs[1] / n, costs[2] / n, costs[3] / n

    def _estep_region(self, region_id):
        b = self._whole_block(region_id, "its E-step")
        b.emission(self.means_, self._covars_)
        if self.warm_start == "best":
            b.warm_start(self.beta, SLOT_LOCAL, report=False)      # labels_local or the previous result, whichever is lower
        else:
            b.restore_labels(SLOT_LOCAL)                 # init_labels = labels_local[id1:id2]  (:479)
        b.solve_fast(self.beta, **self.solver_opts)
        stats, costs, _ = b.posterior_stats(self.beta, self.estimate_type)
        return stats, costs

    def _estep_lockstep(self, regions):
        """block_threads=0: emission and warm start of every whole block queued on its stream, all solves from this thread,
        round by round as the rounds end (Block.solve_group: block for block the labels of the per-block solves), then the
        statistics"""
        if not self.lockstep:
            return None
        blocks = [self._whole_block(r, "its E-step") for r in regions]
        for b in blocks:
            b.emission(self.means_, self._covars_)
            if self.warm_start == "best":
                b.warm_start(self.beta, SLOT_LOCAL, report=False)
            else:
                b.restore_labels(SLOT_LOCAL)
        Block.solve_group(blocks, self.beta, **self.solver_opts)
        out = {}
        for r, b in zip(regions, blocks):
            stats, costs, _ = b.posterior_stats(self.beta, self.estimate_type)
            out[r] = (stats, costs)
        return out

    def _prepare_next_estep(self):
        """While the host fits the K states (the reference's M-step follows its E-step the same way, base.py:399) the GPU has
        nothing to do: the connected components of every whole block's labelling -- the part of the next solve's component
        pass that needs the labels only -- are queued on the blocks' streams now (phmrf_block_prepare_components; the pass
        checks on the device that the labels are still these, so nothing depends on it but the time)."""
        regions = [r for r in self.my_regions if r in self.blocks]
        if not regions or not self.solver_opts.get("use_components", True):
            return None
        def hint(r):
            fn = getattr(self.blocks[r], "prepare_components", None)      # (a hint: test doubles of Block need not have it)
            if fn is not None:
                fn()
        return self.runner.start(hint, regions)

    def _estep_tiles(self):
        """The E-step of every row tile this rank holds (lockstep rounds with the ranks that hold the other tiles of the same
        blocks; runs on the calling thread while the whole blocks run on the runner's threads).
        -> [(stats, cost numerators, owned nodes)] per local tile"""
        out = []

        best = self.warm_start == "best"

        def prepare(tl):
            if not best:
                tl.b.restore_labels(SLOT_LOCAL)              # init_labels = labels_local[id1:id2]  (:479)
            tl.b.emission(self.means_, self._covars_)

        def finish(tl):
            stats, costs, _ = tl.b.posterior_stats(self.beta, self.estimate_type)
            out.append((stats, costs, tl.own_hi - tl.own_lo))

        if self.conductor.groups:
            self.conductor.solve(self.beta, self.solver_opts, prepare=prepare, finish=finish,
                                 warm_slot=SLOT_LOCAL if best else None)
        return out

    def _predict_posteriors(self, X, len_vec, region_id, m_queue=None):
        """(region_id, stats, labels, pairwise_cost, pairwise_cost_normalize, unary_cost, cost1)  (:297-322)."""
        stats, costs = self._estep_region(region_id)          # (raises for a region cut into row tiles, like b3)
        n = float(len_vec[region_id][0])
        out = (region_id, stats, self.blocks[region_id].get_labels(), costs[0] / n, costs[1] / n, costs[2] / n, costs[3] / n)
        if m_queue is not None:
            m_queue.put(out)
            return True
        return out

    # ---- OU parameters <-> Gaussians -------------------------------------------------------------
    def _ou_param_varied_constraint(self, params_vec):
        """means_[c], _covars_[c] = OU(params_vec[c]) + min_covar*I   (:985-1036)."""
        self.means_, self._covars_ = self.tree.mean_cov(np.asarray(params_vec), self.min_covar)
        return True

    def _check_params(self, params):
        return _mstep.check_params(self.tree, params)

    def _ou_lik_varied_constraint(self, params, state_id):
        obj = _mstep.OUObjective(self.tree, self.stats["post"][state_id], self.stats["obs"][state_id],
                                 self.stats["obs*obs.T"][state_id], self.n_samples, self.lambda_0, self.min_covar)
        if _mstep.check_params(self.tree, params) <= -2:                        # :1042-1047
            params = self.init_ou_params[state_id].copy()
        return obj.value(np.asarray(params, dtype=np.float64))

    def _do_mstep(self, stats):
        """The K states are DEALT to the ranks (state c to rank c mod world; the reference fits them one after the other in
        one process, phylo_hmrf.py:1500-1528): every state draws its restarts from a generator of its own, seeded from the
        fit's generator and the state's index, so what a state's fit returns does not depend on who runs it; one all-reduce
        (each state's slot is written by exactly one rank, the others add zeros) leaves the full result on every rank."""
        self.stats = {k: np.array(v, copy=True) for k, v in stats.items() if k in ("post", "obs", "obs*obs.T")}
        K, P, S = self.n_components, self.n_params, self.n_features
        mine = [c for c in range(K) if c % self.world == self.rank] if self.world > 1 else None
        params, means, covars, lik = _mstep.do_mstep(
            self.tree, self.stats, self.params_vec1, self.init_ou_params, self.n_samples, self.lambda_0,
            self.initial_mode, self.initial_w1, self.initial_w1a, self.initial_w2, self.rng, self.min_covar,
            workers=self.mstep_workers, states=mine)
        packed = np.concatenate([params.ravel(), means.ravel(), covars.ravel(), lik.ravel()])
        if self.world > 1:
            packed = self.reducer.allreduce(packed)
        self.params_vec1 = packed[:K * P].reshape(K, P).copy()
        self.means_ = packed[K * P:K * P + K * S].reshape(K, S).copy()
        self._covars_ = packed[K * P + K * S:K * P + K * S + K * S * S].reshape(K, S, S).copy()
        self.lik = packed[-1]

    def close(self):
        # (the M-step's worker pool is process-wide and forked once, before HIP is up: a second fit in the same process
        #  reuses it, so it is NOT ended here -- the CLI and bench.py end it with mstep.close_pool(), otherwise atexit does)
        self.runner.close()
        for b in self.blocks.values():
            b.close()
        self.blocks = {}
        for grp in self.tile_groups.values():
            for tl in grp.local.values():
                tl.b.close()
        self.tile_groups = {}
