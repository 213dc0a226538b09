"""TEST DOUBLE (tests only): the part of the `Block` interface that `phylo_hmrf_amd/tiles.py` drives -- the solve in pieces,
the pins, the boundary / halo rows -- backed by the NumPy move models of the oracle, so that the LOCKSTEP ORCHESTRATION
(payload packing, who sends which row to whom, the pin parity, the sums every tile decides on, the group transport) can be
exercised on CPU, over gloo, without a GPU.  A round here is one colour-ordered ICM sweep (oracle/mrf_moves.icm_sweep) over the
nodes that are neither pinned nor halo; the schedule stops after a round in which no tile changed anything."""
import numpy as np

from oracle import mrf_moves as M
from oracle import ref_numpy as R
from phylo_hmrf_amd import tiles

PIN = 1.0e9


class FakeTileBlock(object):
    def __init__(self, n, S, K):
        self.n, self.S, self.K = n, S, K
        self.labels = np.zeros(n, dtype=np.int64)
        self.saved = {}
        self.top = self.bottom = False
        self.pinned = np.zeros(n, dtype=bool)
        self.in_solve = False

    # -- set-up ----------------------------------------------------------------------------------
    def set_observations(self, X):
        self.X = np.asarray(X, dtype=np.float64)

    def set_graph(self, edges, w):
        """(the tile's edges come from the BLOCK's edge list, restricted and renumbered by tiles.make_group: the oracle's
        edge builder restates the reference's, which knows square diagonal blocks only)"""
        self.eid = np.int64(edges)
        self.w = np.asarray(w, dtype=np.float64)
        self.g = M.Graph(self.n, self.eid, self.w)

    def set_grid(self, H, W, diagonal, num_neighbor=8):
        self.H, self.W, self.diag = H, W, bool(diagonal)
        self.first = [tiles.row_first(i, W, self.diag) for i in range(H + 1)]
        assert self.first[H] == self.n
        ii = np.repeat(np.arange(H), [self.first[i + 1] - self.first[i] for i in range(H)])
        jj = np.concatenate([np.arange(i if self.diag else 0, W) for i in range(H)])
        self.colours, self.ncol = (ii % 2) * 2 + (jj % 2), 4              # 2 x 2 parity classes: no two neighbours share one

    def set_tile(self, top, bottom, sched_n=0):
        self.top, self.bottom = bool(top), bool(bottom)
        self.own = slice(self.first[1] if top else 0, self.first[self.H - 1] if bottom else self.n)

    def set_logprob(self, lp):
        self.unary = -np.asarray(lp, dtype=np.float64)

    def set_labels(self, labels):
        self.labels = np.asarray(labels).astype(np.int64).copy()

    def get_labels(self):
        return self.labels.astype(np.int32)

    def save_labels(self, slot):
        self.saved[slot] = self.labels.copy()

    def restore_labels(self, slot):
        self.labels = self.saved[slot].copy()

    def close(self):
        pass

    # -- the solve in pieces -----------------------------------------------------------------------
    def solve_begin(self, beta, want_init_energy=False, **opts):
        self.beta, self.in_solve, self.status, self.rounds = float(beta), True, 0, 0

    def tile_pins(self, n_top, n_bottom):
        self.pinned[:] = False
        if self.top and n_top:
            self.pinned[:self.first[n_top]] = True
        if self.bottom and n_bottom:
            self.pinned[self.first[self.H - n_bottom]:] = True

    def _energy_owned(self):
        """unary terms of the owned nodes + every edge held by an owned node as its upper / left end (ids ascending)"""
        l = self.labels
        own = np.zeros(self.n, dtype=bool)
        own[self.own] = True
        eu = float(self.unary[np.arange(self.n), l][own].sum())
        a, b = self.eid[:, 0], self.eid[:, 1]
        ep = float((self.w * (l[a] != l[b]))[own[a]].sum())
        return eu, ep

    def solve_round_launch(self):
        u = self.unary.copy()
        pinned = np.flatnonzero(self.pinned)
        u[pinned] = PIN
        u[pinned, self.labels[pinned]] = self.unary[pinned, self.labels[pinned]]
        before = self.labels.copy()
        M.icm_sweep(self.g, u, self.labels, self.beta, self.colours, self.ncol)
        assert np.array_equal(self.labels[pinned], before[pinned])
        self._changed = int(np.sum(self.labels != before))

    def solve_round_collect(self):
        c = np.zeros(128, dtype=np.uint64)
        c[76] = self._changed
        return c, np.array(self._energy_owned())

    def tile_get_boundary(self, top_len, bottom_len):
        top = self.labels[self.first[1]:self.first[2]].astype(np.uint8) if top_len else None
        bot = self.labels[self.first[self.H - 2]:self.first[self.H - 1]].astype(np.uint8) if bottom_len else None
        assert top is None or top.size == top_len
        assert bot is None or bot.size == bottom_len
        return top, bot

    def tile_put_halo(self, top, bottom):
        if top is not None:
            self.labels[:self.first[1]] = top
        if bottom is not None:
            self.labels[self.first[self.H - 1]:] = bottom

    def solve_round_decide(self, counters, energy):
        self.rounds += 1
        self.last_energy = float(energy[0] + self.beta * energy[1])
        # two quiet rounds in a row (one of each pin parity) end the solve; 40 rounds at most
        quiet = int(counters[76]) == 0
        done = quiet and getattr(self, "_prev_quiet", False)
        self._prev_quiet = quiet
        self.status = 1 if done else (2 if self.rounds >= 40 else 0)
        return self.status

    def solve_end(self, want_result=False):
        self.in_solve = False
        self._prev_quiet = False
        if not want_result:
            return None
        eu, ep = self._energy_owned()
        return dict(energy=eu + self.beta * ep, energy_unary=eu, energy_pair=self.beta * ep, energy_init=float("nan"),
                    rounds=self.rounds, converged=self.status == 1, changed=0)
