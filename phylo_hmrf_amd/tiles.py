"""Row tiles: one syntenic block on several GPUs.

The reference runs one process per syntenic block (base.py:357-362) and can make a block smaller only by the centromere
split into independent pieces (utility.py:381-393).  A whole-genome run on 8 GPUs is then bound by its largest block (chr1 =
14 % of the 50 kb workload).  Here a block that is larger than a rank's share is cut into ROW TILES (csrc/tile.hip,
include/phmrf.h "row tiles"): tile t owns the grid rows [r0, r1) and stores one halo row per neighbouring tile.  The tiles
of a block solve their labelling in LOCKSTEP ROUNDS:

    pins of this round's parity  ->  one round of moves per tile (phmrf_mrf_solve_round_launch / _collect)
    ->  ONE all-reduce per block and round of [change counters | energy | the two boundary label rows] per tile
    ->  every tile: halo rows in (phmrf_block_tile_put_halo), the schedule's decision on the SUMS (_round_decide)

so the tiles take the decisions the unsplit block would take on the same numbers, the energy of the whole block never goes
up (tiles never move two adjacent nodes in the same round), and the only traffic is <= 11 KB per tile and round.  The
transport is torch.distributed (RCCL on the GPUs, gloo in the CPU / one-GPU tests); this module never touches device
memory itself.
"""
import math

import numpy as np


# ---- geometry ---------------------------------------------------------------------------------------------------------
def row_first(i, W, diag):
    """first node of grid row i of a block with W columns (diag: the W x W upper triangle, row i = columns i .. W-1)"""
    i = int(i)
    return i * W - (i * (i - 1)) // 2 if diag else i * W


def rows_nodes(r0, r1, W, diag):
    return row_first(r1, W, diag) - row_first(r0, W, diag)


def row_len(i, W, diag):
    return W - i if diag else W


def node_coords(ids, W, diag):
    """grid (row, column) of node ids of a block with W columns (diag: the W x W upper triangle, row-major)"""
    ids = np.asarray(ids, dtype=np.int64)
    if not diag:
        return np.divmod(ids, W)
    # row i starts at i W - i (i - 1) / 2: the largest i whose start is <= id
    i = np.floor(((2 * W + 1) - np.sqrt(np.maximum((2.0 * W + 1) ** 2 - 8.0 * ids, 0.0))) / 2.0).astype(np.int64)
    for _ in range(2):                                   # (the float root can be one off either way)
        i = np.where(i * W - (i * (i - 1)) // 2 > ids, i - 1, i)
        i = np.where((i + 1) * W - ((i + 1) * i) // 2 <= ids, i + 1, i)
    return i, ids - (i * W - (i * (i - 1)) // 2) + i


def edges_fit_grid(ids, H, W, diag, num_neighbor=8):
    """Does every edge of the list [E, 2] join two grid neighbours of the block (H, W, diag)?  The host-side form of what
    phmrf_block_set_grid checks on the device copy (status 1, 'edge list joins nodes that are not grid neighbours'): asked
    BEFORE a block is cut into row tiles, so that a block whose edge list is not the stencil stays whole and takes the
    whole-block path's general-graph fallback instead of failing inside make_group on some ranks only."""
    ids = np.asarray(ids, dtype=np.int64)
    if ids.ndim != 2 or ids.shape[1] < 2:
        return False
    n = rows_nodes(0, H, W, diag)
    if ids.size and (ids[:, :2].min() < 0 or ids[:, :2].max() >= n):
        return False
    i1, j1 = node_coords(ids[:, 0], W, diag)
    i2, j2 = node_coords(ids[:, 1], W, diag)
    di, dj = np.abs(i1 - i2), np.abs(j1 - j2)
    ok = (di <= 1) & (dj <= 1) & (di + dj > 0)
    if num_neighbor == 4:
        ok &= (di + dj) == 1
    return bool(ok.all())


MIN_TILE_ROWS = 8


def split_rows(H, W, diag, parts):
    """owned row ranges [(r0, r1)] of `parts` tiles with (nearly) equal node counts; every tile has >= MIN_TILE_ROWS rows"""
    parts = int(max(1, min(parts, H // MIN_TILE_ROWS)))
    total = rows_nodes(0, H, W, diag)
    cuts = [0]
    for p in range(1, parts):
        want = total * p / float(parts)
        lo, hi = cuts[-1] + MIN_TILE_ROWS, H - MIN_TILE_ROWS * (parts - p)
        r = lo
        if diag:        # rows_nodes(0, r) = r W - r (r - 1) / 2 >= want
            disc = (W + 0.5) ** 2 - 2.0 * want
            r = int(math.ceil((W + 0.5) - math.sqrt(max(disc, 0.0))))
        else:
            r = int(round(want / float(W)))
        cuts.append(int(min(max(r, lo), hi)))
    cuts.append(H)
    return [(cuts[i], cuts[i + 1]) for i in range(parts)]


def plan(blocks, world, split_above=1.0, force_parts=None):
    """Units of work of a block list [(H, W, diag)] on `world` ranks: whole blocks, and row tiles of the blocks that hold more
    than split_above x (all nodes / world).  -> list of dicts {block, tile, ntiles, r0, r1, nodes} (block order, tiles top
    to bottom).  force_parts {block index: parts} overrides the rule (tests)."""
    sizes = [rows_nodes(0, H, W, d) for H, W, d in blocks]
    share = sum(sizes) / float(max(world, 1))
    units = []
    for bi, (H, W, d) in enumerate(blocks):
        parts = 1
        if force_parts and bi in force_parts:
            parts = int(force_parts[bi])
        elif world > 1 and sizes[bi] > split_above * share:
            parts = int(math.ceil(sizes[bi] / (split_above * share)))
        rows = split_rows(H, W, d, parts) if parts > 1 else [(0, H)]
        for t, (r0, r1) in enumerate(rows):
            units.append(dict(block=bi, tile=t, ntiles=len(rows), r0=r0, r1=r1, nodes=rows_nodes(r0, r1, W, d)))
    return units


def assign(units, world):
    """longest-processing-time-first over the units -> owner rank per unit (deterministic)"""
    from .dist import lpt_assign
    return [int(o) for o in lpt_assign([u["nodes"] for u in units], world)]


# ---- one tile ---------------------------------------------------------------------------------------------------------
class Tile(object):
    """Rows [r0, r1) of the block (H, W, diag) plus a halo row towards each neighbouring tile, as a Block of its own."""

    def __init__(self, H, W, diag, r0, r1, tile, ntiles, S, K, factory):
        self.H, self.W, self.diag = int(H), int(W), bool(diag)
        self.r0, self.r1, self.tile, self.ntiles = int(r0), int(r1), int(tile), int(ntiles)
        self.top, self.bottom = tile > 0, tile < ntiles - 1
        self.s0, self.s1 = self.r0 - int(self.top), self.r1 + int(self.bottom)            # stored rows
        self.Hl = self.s1 - self.s0
        self.Wl = self.W - self.s0 if self.diag else self.W
        self.node0 = row_first(self.s0, self.W, self.diag)                               # global id of the first stored node
        self.n = rows_nodes(self.s0, self.s1, self.W, self.diag)
        self.own_lo = rows_nodes(self.s0, self.r0, self.W, self.diag)                    # local ids of the owned nodes
        self.own_hi = rows_nodes(self.s0, self.r1, self.W, self.diag)
        self.n_block = rows_nodes(0, self.H, self.W, self.diag)
        # lengths of the rows that travel: my first / last owned row (out), the neighbours' last / first owned row (in)
        self.top_out = row_len(self.r0, self.W, self.diag) if self.top else 0
        self.bot_out = row_len(self.r1 - 1, self.W, self.diag) if self.bottom else 0
        self.top_in = row_len(self.r0 - 1, self.W, self.diag) if self.top else 0
        self.bot_in = row_len(self.r1, self.W, self.diag) if self.bottom else 0
        self.b = factory(self.n, S, K)

    def geometry(self):
        """(H, W, diagonal) of the tile's own block"""
        return self.Hl, self.Wl, self.diag

    def configure(self):
        """after the graph / grid is set: tell the library which rows are owned"""
        self.b.set_tile(self.top, self.bottom, self.n_block)

    def global_slice(self):
        """the stored nodes as a slice of the whole block's node ids"""
        return slice(self.node0, self.node0 + self.n)

    def owned_global_slice(self):
        return slice(self.node0 + self.own_lo, self.node0 + self.own_hi)

    def owned_local_slice(self):
        return slice(self.own_lo, self.own_hi)


# ---- the tiles of one block ---------------------------------------------------------------------------------------------
N_COUNTERS = 128


class TileGroup(object):
    """The tiles of ONE block: those this rank holds (`local`: {tile index: Tile}) and the transport to the ranks that hold
    the others.  `comm.allreduce_i64(np.int64 array) -> summed array` (identity when every tile is local)."""

    def __init__(self, block_id, geometry, ntiles, local, comm=None):
        self.block_id = block_id
        self.H, self.W, self.diag = geometry
        self.ntiles = int(ntiles)
        self.local = dict(local)
        self.comm = comm
        self.row_words = (row_len(0, self.W, self.diag) + 7) // 8          # int64 words that hold the longest row
        self.slot = N_COUNTERS + 3 + 2 * self.row_words
        self.rounds = 0
        self.status = 0
        self.missing = set(range(self.ntiles)) - set(self.local) if comm is None else set()
        self.result = None
        # host wall time of the rounds' parts since reset_clock(): waiting for the local tiles' kernels (collect), the
        # exchange proper (pack, all-reduce over the holders, unpack + halo rows in), the schedule's decision
        self.clock = dict(rounds=0, collect=0.0, exchange=0.0, allreduce=0.0, decide=0.0)

    def reset_clock(self):
        for k in self.clock:
            self.clock[k] = 0 if k == "rounds" else 0.0

    # the payload of tile t: [counters 128 | energy 2 (float64 bits) | 1 + the status its holder decided last round |
    # first owned row | last owned row], int64 words
    def _pack(self, buf, t, counters, energy, top, bot):
        o = t * self.slot
        buf[o:o + N_COUNTERS] = counters.view(np.int64)
        buf[o + N_COUNTERS:o + N_COUNTERS + 2] = np.asarray(energy, dtype=np.float64).view(np.int64)
        buf[o + N_COUNTERS + 2] = 1 + self.status
        rows = buf[o + N_COUNTERS + 3:o + self.slot].view(np.uint8)
        if top is not None:
            rows[:top.size] = top
        if bot is not None:
            rows[8 * self.row_words:8 * self.row_words + bot.size] = bot

    def _rows(self, buf, t):
        o = t * self.slot + N_COUNTERS + 3
        rows = buf[o:o + 2 * self.row_words].view(np.uint8)
        return rows[:8 * self.row_words], rows[8 * self.row_words:]

    def sync_halos(self):
        """halo rows <- the neighbours' boundary rows (after labels were set tile by tile; a solve keeps them current)"""
        buf = np.zeros(self.ntiles * self.slot, dtype=np.int64)
        for t in sorted(self.local):
            tl = self.local[t]
            top, bot = tl.b.tile_get_boundary(tl.top_out, tl.bot_out)
            self._pack(buf, t, np.zeros(N_COUNTERS, dtype=np.uint64), np.zeros(2), top, bot)
        if self.comm is not None:
            buf = self.comm.allreduce_i64(buf)
        for t in sorted(self.local):
            tl = self.local[t]
            halo_top = self._rows(buf, t - 1)[1][:tl.top_in] if (tl.top and (t - 1) not in self.missing) else None
            halo_bot = self._rows(buf, t + 1)[0][:tl.bot_in] if (tl.bottom and (t + 1) not in self.missing) else None
            tl.b.tile_put_halo(halo_top, halo_bot)

    def choose_start(self, beta, slot):
        """Block.warm_start for the tiles of one block: the current labels (the previous E-step's) or the snapshot in `slot`
        (labels_local), whichever has the lower energy summed over the tiles' owned rows -- one decision for all of them, so
        the halo rows stay what the neighbours hold.  -> took the snapshot"""
        e = np.zeros(2 * self.ntiles, dtype=np.float64)
        for t in sorted(self.local):
            e[2 * t], e[2 * t + 1], _ = self.local[t].b.warm_start(beta, slot, choose=False)
        if self.comm is not None:
            e = self.comm.allreduce_i64(e.view(np.int64)).view(np.float64)
        e = e.reshape(self.ntiles, 2)
        take_saved = not (float(e[:, 0].sum()) < float(e[:, 1].sum()))
        if take_saved:
            for t in sorted(self.local):
                self.local[t].b.restore_labels(slot)
        return take_saved

    def begin(self, beta, opts):
        self.rounds = 0
        self.status = 0
        self.result = None
        for t in sorted(self.local):
            self.local[t].b.solve_begin(beta, **opts)

    def launch(self):
        """pins of this round's parity, then one round of moves of every local tile (asynchronous)"""
        phase = self.rounds & 1
        for t in sorted(self.local):
            b = self.local[t].b
            b.tile_pins(1 + phase, 2 - phase)
            b.solve_round_launch()

    def finish_round(self):
        """wait for the local tiles, exchange with the others, decide.  -> status (the same on every rank)"""
        import time
        buf = np.zeros(self.ntiles * self.slot, dtype=np.int64)
        t0 = time.perf_counter()
        got = []
        for t in sorted(self.local):
            tl = self.local[t]
            counters, energy = tl.b.solve_round_collect()
            got.append((t, tl, counters, energy))
        t1 = time.perf_counter()
        for t, tl, counters, energy in got:
            top, bot = tl.b.tile_get_boundary(tl.top_out, tl.bot_out)
            self._pack(buf, t, counters, energy, top, bot)
        ta = time.perf_counter()
        if self.comm is not None:
            buf = self.comm.allreduce_i64(buf)
        tb = time.perf_counter()
        tot_c = np.zeros(N_COUNTERS, dtype=np.uint64)
        tot_e = np.zeros(2, dtype=np.float64)
        seen = set()
        for t in range(self.ntiles):                       # fixed order: every rank forms the same sums
            o = t * self.slot
            tot_c += buf[o:o + N_COUNTERS].view(np.uint64)
            tot_e += buf[o + N_COUNTERS:o + N_COUNTERS + 2].view(np.float64)
            if t not in self.missing:
                seen.add(int(buf[o + N_COUNTERS + 2]) - 1)
        # every holder decides on the same sums and so reaches the same status; one that did not (a failed tile, a rank on
        # another code path) would otherwise hang its partners in the next all-reduce -- fail here instead, on every rank
        if len(seen) > 1:
            raise RuntimeError("block %d: the ranks that hold its row tiles disagree on the schedule (status of the previous "
                               "round: %s)" % (self.block_id, sorted(seen)))
        status = None
        for t in sorted(self.local):
            tl = self.local[t]
            halo_top = halo_bot = None
            if tl.top and (t - 1) not in self.missing:
                halo_top = self._rows(buf, t - 1)[1][:tl.top_in]           # the upper neighbour's last owned row
            if tl.bottom and (t + 1) not in self.missing:
                halo_bot = self._rows(buf, t + 1)[0][:tl.bot_in]           # the lower neighbour's first owned row
            tl.b.tile_put_halo(halo_top, halo_bot)
        t2 = time.perf_counter()
        for t in sorted(self.local):
            st = self.local[t].b.solve_round_decide(tot_c, tot_e)
            assert status is None or status == st, "tiles of one block disagree on the schedule"
            status = st
        t3 = time.perf_counter()
        c = self.clock
        c["rounds"] += 1
        c["collect"] += t1 - t0
        c["exchange"] += t2 - t1
        c["allreduce"] += tb - ta
        c["decide"] += t3 - t2
        self.rounds += 1
        self.status = status
        return status

    def end(self, want_result=False):
        res = []
        for t in sorted(self.local):
            b = self.local[t].b
            b.tile_pins(0, 0)
            res.append(b.solve_end(want_result))
        if want_result:
            e = np.zeros(4 * self.ntiles, dtype=np.float64)
            for t, r in zip(sorted(self.local), res):
                e[4 * t:4 * t + 4] = [r["energy"], r["energy_unary"], r["energy_pair"], float(r["changed"])]
            if self.comm is not None:
                e = self.comm.allreduce_i64(e.view(np.int64)).view(np.float64)
            e = e.reshape(self.ntiles, 4)
            self.result = dict(energy=float(e[:, 0].sum()), energy_unary=float(e[:, 1].sum()), energy_pair=float(e[:, 2].sum()),
                               rounds=res[0]["rounds"], converged=res[0]["converged"], changed=int(res[0]["changed"]))
        return self.result


class Conductor(object):
    """Runs the lockstep solves of every split block this rank has tiles of, all at once: per super-round the rounds of all
    unfinished groups are queued first (the tiles' kernels overlap on their streams), then finished group by group in block
    order -- the same order on every rank, so the groups' collectives never wait on each other in a cycle."""

    def __init__(self, groups):
        self.groups = sorted(groups, key=lambda g: g.block_id)

    def solve(self, beta, opts, prepare=None, finish=None, want_result=False, warm_slot=None):
        """prepare(tile) before the solve (restore labels, emission), finish(tile) after it (posteriors / statistics);
        warm_slot: after prepare, every group starts from its current labels or from that snapshot, whichever is lower"""
        for g in self.groups:
            for t in sorted(g.local):
                if prepare is not None:
                    prepare(g.local[t])
            if warm_slot is not None:
                g.choose_start(beta, warm_slot)
            g.begin(beta, opts)
        active = list(self.groups)
        while active:
            for g in active:
                g.launch()
            still = []
            for g in active:
                if g.finish_round() == 0:
                    still.append(g)
                else:
                    g.end(want_result)
                    if finish is not None:
                        for t in sorted(g.local):
                            finish(g.local[t])
            active = still
        return [g.result for g in self.groups]


# ---- transport ---------------------------------------------------------------------------------------------------------
class GroupComm(object):
    """all-reduce(sum) of an int64 vector over the ranks that hold tiles of one block (a torch.distributed group).  With the
    nccl backend (RCCL) the vector travels through a device tensor, with gloo through host memory."""

    def __init__(self, ranks, device=None):
        import os
        import torch.distributed as dist
        self.ranks = sorted(set(int(r) for r in ranks))
        # PHMRF_TILE_BACKEND=gloo: the tiles' control messages (<= 11 KB per tile and round, host-resident on both ends)
        # over a gloo group next to an RCCL world -- an operator's escape hatch; the default is the world's backend
        backend = os.environ.get("PHMRF_TILE_BACKEND") or None
        if backend == "gloo":
            device = None
        self.group = dist.new_group(ranks=self.ranks, backend=backend) if len(self.ranks) > 1 else None   # (every rank must call this)
        self.device = device
        self._buf = None

    def allreduce_i64(self, vec):
        if self.group is None:
            return vec
        import torch
        import torch.distributed as dist
        if self._buf is None or self._buf.numel() != vec.size:
            self._buf = torch.zeros(vec.size, dtype=torch.int64, device=self.device or "cpu")
        self._buf.copy_(torch.from_numpy(vec))
        dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.group)
        return self._buf.cpu().numpy().copy()


# ---- building the local tiles of a block ---------------------------------------------------------------------------------
def make_group(block_id, geometry, rows, owners, rank, S, K, factory, load, comm=None, num_neighbor=8, beta1=0.5, edges=None):
    """The TileGroup of one split block on this rank.  rows = [(r0, r1)] of all its tiles, owners[t] = rank of tile t (None:
    held by nobody -- a one-GPU rehearsal of one rank's share).  load(tile) must put the observations of the tile's stored
    nodes (tile.global_slice() of the block) into tile.b; the graph is then built on the device from them -- or, with
    `edges` (the block's host edge list [E, 3]: id1, id2, distance, as the reference builds it, utility.py:1955-1960), from
    the edges whose two ends the tile stores."""
    H, W, diag = geometry
    local = {}
    for t, (r0, r1) in enumerate(rows):
        if owners[t] != rank:
            continue
        tl = Tile(H, W, diag, r0, r1, t, len(rows), S, K, factory)
        load(tl)
        if edges is None:
            tl.b.build_grid_graph(tl.Hl, tl.Wl, tl.diag, num_neighbor, beta1)
        else:
            e = np.asarray(edges)
            ids = np.int64(e[:, 0:2])
            keep = (ids.min(axis=1) >= tl.node0) & (ids.max(axis=1) < tl.node0 + tl.n)
            tl.b.set_graph(ids[keep] - tl.node0, np.exp(-beta1 * e[keep, 2]))
            tl.b.set_grid(tl.Hl, tl.Wl, tl.diag, num_neighbor)
        tl.configure()
        local[t] = tl
    return TileGroup(block_id, geometry, len(rows), local, comm)
