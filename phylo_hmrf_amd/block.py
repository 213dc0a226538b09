"""`Block`: one syntenic block resident on one MI355X, a thin object wrapper over the C ABI."""
import ctypes

import numpy as np

from . import _lib
from ._lib import SolveOpts, SolveResult, as_f64, check, ptr_d, ptr_i32, ptr_i64


class Block(object):
    """Device-resident state of one syntenic block (reference: one forked process per block,
    base.py:357-362).  X, the neighbour graph, logprob[n,K] and the labels live in HBM."""

    def __init__(self, n, S, K):
        _lib.require_gpu()
        self._L = _lib.load()
        self.n, self.S, self.K = int(n), int(S), int(K)
        h = ctypes.c_void_p()
        check(self._L.phmrf_block_create(self.n, self.S, self.K, ctypes.byref(h)))
        self._h = h

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.phmrf_block_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self._L.phmrf_block_sync(self._h))

    def set_stream(self, stream_ptr):
        check(self._L.phmrf_block_set_stream(self._h, ctypes.c_void_p(stream_ptr)))

    # -- inputs -----------------------------------------------------------------------------------
    def set_observations(self, X):
        X = as_f64(X)
        assert X.shape == (self.n, self.S), (X.shape, self.n, self.S)
        check(self._L.phmrf_block_set_observations(self._h, ptr_d(X)))

    def set_observations_dev(self, dev_ptr):
        check(self._L.phmrf_block_set_observations_dev(self._h, ctypes.c_void_p(dev_ptr)))

    def set_graph(self, edges, w):
        edges = np.ascontiguousarray(np.asarray(edges)[:, 0:2], dtype=np.int64)
        w = as_f64(w)
        assert edges.shape[0] == w.shape[0]
        check(self._L.phmrf_block_set_graph(self._h, edges.shape[0], ptr_i64(edges), ptr_d(w)))

    def set_grid(self, H, W, diagonal, num_neighbor=8):
        check(self._L.phmrf_block_set_grid(self._h, int(H), int(W), int(bool(diagonal)), int(num_neighbor)))

    def build_grid_graph(self, H, W, diagonal, num_neighbor=8, beta1=0.5):
        """Graph built on the device from the resident observations (no host edge list)."""
        check(self._L.phmrf_block_build_grid_graph(self._h, int(H), int(W), int(bool(diagonal)), int(num_neighbor),
                                                   float(beta1)))

    def get_adjacency(self):
        D = ctypes.c_int(0)
        check(self._L.phmrf_block_get_adjacency(self._h, ctypes.byref(D), None, None))
        nbr = np.empty((self.n, D.value), dtype=np.int32)
        wgt = np.empty((self.n, D.value), dtype=np.float32)
        check(self._L.phmrf_block_get_adjacency(self._h, ctypes.byref(D), ptr_i32(nbr),
                                                wgt.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        return nbr, wgt

    def set_labels(self, labels):
        lab = np.ascontiguousarray(np.asarray(labels), dtype=np.int32)   # float labels are cast (base.py:381,394)
        assert lab.shape == (self.n,)
        check(self._L.phmrf_block_set_labels(self._h, ptr_i32(lab)))

    def get_labels(self):
        out = np.empty(self.n, dtype=np.int32)
        check(self._L.phmrf_block_get_labels(self._h, ptr_i32(out)))
        return out

    def save_labels(self, slot):
        check(self._L.phmrf_block_save_labels(self._h, slot))

    def restore_labels(self, slot):
        check(self._L.phmrf_block_restore_labels(self._h, slot))

    def warm_start(self, beta, slot, choose=True, report=True):
        """current labels (the previous E-step's) or the snapshot in `slot` (labels_local), whichever has the lower energy
        under the resident logprob -> (e_current, e_saved, took_saved); choose=False: only the energies.  report=False: the
        two evaluations and the choice are queued on the block's stream and the call returns at once (-> None)"""
        if not report:
            check(self._L.phmrf_block_warm_start(self._h, float(beta), int(slot), int(bool(choose)), None, None, None))
            return None
        ec, es, took = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int(0)
        check(self._L.phmrf_block_warm_start(self._h, float(beta), int(slot), int(bool(choose)), ctypes.byref(ec),
                                             ctypes.byref(es), ctypes.byref(took)))
        return ec.value, es.value, bool(took.value)

    def get_saved_labels(self, slot):
        out = np.empty(self.n, dtype=np.int32)
        check(self._L.phmrf_block_get_saved_labels(self._h, slot, ptr_i32(out)))
        return out

    # -- b1 ---------------------------------------------------------------------------------------
    def emission(self, means, covars):
        means, covars = as_f64(means), as_f64(covars)
        assert means.shape == (self.K, self.S) and covars.shape == (self.K, self.S, self.S)
        check(self._L.phmrf_emission(self._h, ptr_d(means), ptr_d(covars)))

    def get_logprob(self):
        out = np.empty((self.n, self.K), dtype=np.float64)
        check(self._L.phmrf_block_get_logprob(self._h, ptr_d(out)))
        return out

    def set_logprob(self, logprob):
        lp = as_f64(logprob)
        assert lp.shape == (self.n, self.K)
        check(self._L.phmrf_block_set_logprob(self._h, ptr_d(lp)))

    # -- b2 ---------------------------------------------------------------------------------------
    def solve(self, beta, max_rounds=64, use_chains=True, use_components=True, init_mode=0, use_strips=True,
              use_expansion=True, min_changed=0, energy_tol_ppb=0, use_coarse=True, coarse_start=0):
        o = SolveOpts(int(max_rounds), int(use_chains), int(use_components), int(init_mode), int(use_strips),
                      int(use_expansion), int(min_changed), int(use_coarse), int(energy_tol_ppb), int(coarse_start))
        r = SolveResult()
        check(self._L.phmrf_mrf_solve(self._h, float(beta), ctypes.byref(o), ctypes.byref(r)))
        return dict(energy=r.energy, energy_unary=r.energy_unary, energy_pair=r.energy_pair,
                    energy_init=r.energy_init, rounds=r.rounds, converged=bool(r.converged), changed=r.changed)

    def solve_fast(self, beta, max_rounds=64, use_chains=True, use_components=True, init_mode=0, use_strips=True,
                   use_expansion=True, min_changed=0, energy_tol_ppb=0, use_coarse=True, coarse_start=0):
        """Same without the two energy evaluations."""
        o = SolveOpts(int(max_rounds), int(use_chains), int(use_components), int(init_mode), int(use_strips),
                      int(use_expansion), int(min_changed), int(use_coarse), int(energy_tol_ppb), int(coarse_start))
        check(self._L.phmrf_mrf_solve(self._h, float(beta), ctypes.byref(o), None))

    @staticmethod
    def solve_group(blocks, beta, max_rounds=64, use_chains=True, use_components=True, init_mode=0, use_strips=True,
                    use_expansion=True, min_changed=0, energy_tol_ppb=0, use_coarse=True, coarse_start=0):
        """solve_fast of several independent blocks from THIS thread, round by round as the rounds end (phmrf_mrf_solve_group): their kernels
        overlap on the GPU as with one host thread per block, without the threads.  List the largest blocks first."""
        if not blocks:
            return
        o = SolveOpts(int(max_rounds), int(use_chains), int(use_components), int(init_mode), int(use_strips),
                      int(use_expansion), int(min_changed), int(use_coarse), int(energy_tol_ppb), int(coarse_start))
        arr = (ctypes.c_void_p * len(blocks))(*[b._h for b in blocks])
        check(blocks[0]._L.phmrf_mrf_solve_group(arr, len(blocks), float(beta), ctypes.byref(o)))

    # -- the solve in pieces (lockstep rounds of the row tiles of one block: tiles.py) ----------------
    def solve_begin(self, beta, want_init_energy=False, max_rounds=64, use_chains=True, use_components=True, init_mode=0,
                    use_strips=True, use_expansion=True, min_changed=0, energy_tol_ppb=0, use_coarse=True, coarse_start=0):
        o = SolveOpts(int(max_rounds), int(use_chains), int(use_components), int(init_mode), int(use_strips),
                      int(use_expansion), int(min_changed), int(use_coarse), int(energy_tol_ppb), int(coarse_start))
        check(self._L.phmrf_mrf_solve_begin(self._h, float(beta), ctypes.byref(o), int(bool(want_init_energy))))

    def solve_round_launch(self):
        check(self._L.phmrf_mrf_solve_round_launch(self._h))

    def solve_round_collect(self):
        """-> (counters uint64[128], energy float64[2] = unary sum, pair sum without beta) of this block after the round"""
        c = np.zeros(128, dtype=np.uint64)
        e = np.zeros(2, dtype=np.float64)
        check(self._L.phmrf_mrf_solve_round_collect(self._h, c.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), ptr_d(e)))
        return c, e

    def solve_round_decide(self, counters, energy):
        """-> 0 another round, 1 converged, 2 stopped by max_rounds / the launch budget"""
        c = np.ascontiguousarray(counters, dtype=np.uint64)
        e = np.ascontiguousarray(energy, dtype=np.float64)
        st = ctypes.c_int(0)
        check(self._L.phmrf_mrf_solve_round_decide(self._h, c.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), ptr_d(e),
                                                   ctypes.byref(st)))
        return st.value

    def solve_end(self, want_result=False):
        if not want_result:
            check(self._L.phmrf_mrf_solve_end(self._h, None))
            return None
        r = SolveResult()
        check(self._L.phmrf_mrf_solve_end(self._h, ctypes.byref(r)))
        return dict(energy=r.energy, energy_unary=r.energy_unary, energy_pair=r.energy_pair,
                    energy_init=r.energy_init, rounds=r.rounds, converged=bool(r.converged), changed=r.changed)

    # -- row tiles ---------------------------------------------------------------------------------
    def set_tile(self, top, bottom, sched_n=0):
        check(self._L.phmrf_block_set_tile(self._h, int(bool(top)), int(bool(bottom)), int(sched_n)))

    def tile_pins(self, n_top, n_bottom):
        check(self._L.phmrf_block_tile_pins(self._h, int(n_top), int(n_bottom)))

    def tile_get_boundary(self, top_len, bottom_len):
        """-> (first owned row or None, last owned row or None) as uint8 arrays of the given lengths (0: no such cut)"""
        u8 = ctypes.POINTER(ctypes.c_uint8)
        top = np.empty(top_len, dtype=np.uint8) if top_len else None
        bot = np.empty(bottom_len, dtype=np.uint8) if bottom_len else None
        check(self._L.phmrf_block_tile_get_boundary(self._h, top.ctypes.data_as(u8) if top_len else None,
                                                    bot.ctypes.data_as(u8) if bottom_len else None))
        return top, bot

    def tile_put_halo(self, top, bottom):
        u8 = ctypes.POINTER(ctypes.c_uint8)
        t = None if top is None else np.ascontiguousarray(top, dtype=np.uint8)
        b = None if bottom is None else np.ascontiguousarray(bottom, dtype=np.uint8)
        check(self._L.phmrf_block_tile_put_halo(self._h, None if t is None else t.ctypes.data_as(u8),
                                                None if b is None else b.ctypes.data_as(u8)))

    def icm_sweep(self, beta):
        c = ctypes.c_int64(0)
        check(self._L.phmrf_mrf_icm_sweep(self._h, float(beta), ctypes.byref(c)))
        return c.value

    def chain_sweep(self, beta, family):
        c = ctypes.c_int64(0)
        check(self._L.phmrf_mrf_chain_sweep(self._h, float(beta), int(family), ctypes.byref(c)))
        return c.value

    def prepare_components(self):
        """Queue the labels-only part of the next component pass (connected components of equal label) on the block's
        stream -- between two E-steps, while the host runs the M-step.  A hint: the pass checks on the device that the labels
        are still the ones prepared for, so results never depend on it."""
        check(self._L.phmrf_block_prepare_components(self._h))

    def component_pass(self, beta):
        c = ctypes.c_int64(0)
        check(self._L.phmrf_mrf_component_pass(self._h, float(beta), ctypes.byref(c)))
        return c.value

    def graph_expansion(self, beta, alpha):
        """one alpha-expansion of the whole graph by an exact minimum cut (general graphs; include/phmrf.h) -> labels changed"""
        c = ctypes.c_int64(0)
        check(self._L.phmrf_mrf_graph_expansion(self._h, float(beta), int(alpha), ctypes.byref(c)))
        return c.value

    def strip_pass(self, beta, orient, shift_r, shift_c, alpha=-1):
        c = ctypes.c_int64(0)
        check(self._L.phmrf_mrf_strip_pass(self._h, float(beta), int(orient), int(shift_r), int(shift_c), int(alpha),
                                           ctypes.byref(c)))
        return c.value

    def strip_multi_pass(self, beta, orient, shift_r, shift_c, labels=None):
        """the strip alpha-expansions of `labels` (default: all K) on one cut, ascending, in one launch"""
        mask = 0
        for a in (range(self.K) if labels is None else labels):
            mask |= 1 << int(a)
        c = ctypes.c_int64(0)
        check(self._L.phmrf_mrf_strip_multi_pass(self._h, float(beta), int(orient), int(shift_r), int(shift_c),
                                                 ctypes.c_uint64(mask), ctypes.byref(c)))
        return c.value

    def coarse_pass(self, beta, scale, offset, alpha, shift_r=0, shift_c=0):
        c = ctypes.c_int64(0)
        check(self._L.phmrf_mrf_coarse_pass(self._h, float(beta), int(scale), int(offset), int(alpha), int(shift_r),
                                            int(shift_c), ctypes.byref(c)))
        return c.value

    def coarse_problem(self, beta, scale, offset, alpha):
        """-> (D[nc], lam[nc,4]) of the super-cell problem (tests)."""
        nc = ctypes.c_int64(0)
        check(self._L.phmrf_block_coarse_problem(self._h, float(beta), int(scale), int(offset), int(alpha), ctypes.byref(nc),
                                                 None, None))
        D = np.zeros(nc.value, dtype=np.float32)
        lam = np.zeros((nc.value, 4), dtype=np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        check(self._L.phmrf_block_coarse_problem(self._h, float(beta), int(scale), int(offset), int(alpha), ctypes.byref(nc),
                                                 D.ctypes.data_as(fp), lam.ctypes.data_as(fp)))
        return D, lam

    def energy(self, beta):
        e, eu, ep = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        check(self._L.phmrf_mrf_energy(self._h, float(beta), ctypes.byref(e), ctypes.byref(eu), ctypes.byref(ep)))
        return e.value, eu.value, ep.value

    # -- b3 ---------------------------------------------------------------------------------------
    def n_stats(self):
        return self.K * (1 + self.S + self.S * self.S)

    def posterior_stats(self, beta, estimate_type, want_posteriors=False):
        """-> (stats dict with the reference's keys, costs[4] un-normalised sums, posteriors or None)."""
        K, S = self.K, self.S
        st = np.zeros(self.n_stats(), dtype=np.float64)
        costs = np.zeros(4, dtype=np.float64)
        post = np.empty((self.n, K), dtype=np.float64) if want_posteriors else None
        check(self._L.phmrf_posterior_stats(self._h, float(beta), int(estimate_type), ptr_d(st), ptr_d(costs),
                                            ptr_d(post) if want_posteriors else None))
        return unpack_stats(st, K, S), costs, post

    def posterior_stats_dev(self, beta, estimate_type, out_dev_ptr):
        check(self._L.phmrf_posterior_stats_dev(self._h, float(beta), int(estimate_type), ctypes.c_void_p(out_dev_ptr)))

    # -- initialisation ---------------------------------------------------------------------------
    def kmeans_step(self, centers, write_labels=False):
        """One Lloyd step on the device: -> (sums[K,S], counts[K], inertia)."""
        c = as_f64(centers)
        assert c.shape == (self.K, self.S)
        out = np.zeros(self.K * self.S + self.K + 1)
        check(self._L.phmrf_kmeans_step(self._h, ptr_d(c), int(bool(write_labels)), ptr_d(out)))
        KS = self.K * self.S
        return out[:KS].reshape(self.K, self.S), out[KS:KS + self.K], float(out[KS + self.K])

    def kmeans_moments(self, centers, write_labels=False):
        """The Lloyd step with the per-cluster second moments: -> (sums[K,S], counts[K], inertia, outer[K,S,S])."""
        c = as_f64(centers)
        assert c.shape == (self.K, self.S)
        K, S = self.K, self.S
        out = np.zeros(K * S + K + 1 + K * S * S)
        check(self._L.phmrf_kmeans_moments(self._h, ptr_d(c), int(bool(write_labels)), ptr_d(out)))
        KS = K * S
        return (out[:KS].reshape(K, S), out[KS:KS + K], float(out[KS + K]), out[KS + K + 1:].reshape(K, S, S))

    # -- timing -----------------------------------------------------------------------------------
    def enable_timing(self, on=True, classes=None):
        """classes: names of the kernel classes that get event pairs (None: all)"""
        mask = 0xFFFFFFFF
        if classes is not None:
            mask = 0
            for c in classes:
                mask |= 1 << _lib.KERNEL_CLASSES.index(c)
        check(self._L.phmrf_block_set_timing_classes(self._h, ctypes.c_uint32(mask)))
        check(self._L.phmrf_block_enable_timing(self._h, int(on)))

    def reset_timing(self):
        check(self._L.phmrf_block_reset_timing(self._h))

    def timing(self):
        ms = (ctypes.c_double * _lib.NUM_KERNEL_CLASSES)()
        ln = (ctypes.c_int64 * _lib.NUM_KERNEL_CLASSES)()
        check(self._L.phmrf_block_get_timing(self._h, _lib.NUM_KERNEL_CLASSES, ms, ln))
        return {name: (ms[i], ln[i]) for i, name in enumerate(_lib.KERNEL_CLASSES)}

    def timing_first(self):
        """The part of timing() that belongs to launches of the first round of a solve (a warm start's full sweeps)."""
        ms = (ctypes.c_double * _lib.NUM_KERNEL_CLASSES)()
        ln = (ctypes.c_int64 * _lib.NUM_KERNEL_CLASSES)()
        check(self._L.phmrf_block_get_timing_first(self._h, _lib.NUM_KERNEL_CLASSES, ms, ln))
        return {name: (ms[i], ln[i]) for i, name in enumerate(_lib.KERNEL_CLASSES)}

    _WORK_KEYS = ("units", "cells", "staged_cells", "dp_steps", "launches", "swept_cells", "label_cells", "proposal_nodes",
                  "mask_label_cells", "mask_strip_cells")

    def work(self):
        """Device-counted work of the strip kernels since reset_timing (include/phmrf.h: phmrf_block_get_work_ex)."""
        out = (ctypes.c_int64 * 10)()
        check(self._L.phmrf_block_get_work_ex(self._h, 0, 10, out))
        return dict(zip(self._WORK_KEYS, out))

    def work_first(self):
        """The part of work() done in the first round of every solve (a warm start's full sweeps) since reset_timing."""
        out = (ctypes.c_int64 * 10)()
        check(self._L.phmrf_block_get_work_ex(self._h, 1, 10, out))
        return dict(zip(self._WORK_KEYS, out))

    def intervals(self, kernel_class):
        """[start_ms, end_ms] of every timed interval of the class, on the device's common time base -> array [m,2]."""
        k = _lib.KERNEL_CLASSES.index(kernel_class)
        cnt = ctypes.c_int64(0)
        check(self._L.phmrf_block_get_intervals(self._h, k, None, 0, ctypes.byref(cnt)))
        out = np.zeros((cnt.value, 2), dtype=np.float64)
        if cnt.value:
            check(self._L.phmrf_block_get_intervals(self._h, k, ptr_d(out), cnt.value, ctypes.byref(cnt)))
        return out


def time_base_reset():
    """t = 0 of the interval time line of the current device (all blocks)."""
    check(_lib.load().phmrf_time_base_reset())


def unpack_stats(vec, K, S):
    """flat [K + K*S + K*S*S] -> the reference's stats dict (phylo_hmrf.py:311-314)."""
    vec = np.asarray(vec, dtype=np.float64)
    return {"post": vec[:K].copy(),
            "obs": vec[K:K + K * S].reshape(K, S).copy(),
            "obs*obs.T": vec[K + K * S:K + K * S + K * S * S].reshape(K, S, S).copy()}


def pack_stats(stats):
    return np.concatenate([np.ravel(stats["post"]), np.ravel(stats["obs"]), np.ravel(stats["obs*obs.T"])])
