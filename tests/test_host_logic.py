"""CPU: host-side logic that needs no device -- grid inference of the pygco drop-in, block lists of the workloads,
sharding arithmetic."""
import numpy as np
import pytest

from oracle import ref_numpy as R


@pytest.mark.parametrize("H,W,diag", [(9, 9, True), (7, 10, False), (1, 12, False), (12, 1, False), (3, 3, True),
                                      (2, 3, False), (2, 2, True), (40, 50, False)])
@pytest.mark.parametrize("nn", [8, 4])
def test_grid_candidates_contain_the_true_geometry(H, W, diag, nn):
    from phylo_hmrf_amd.pygco_compat import grid_candidates
    n = H * (H + 1) // 2 if diag else H * W
    X = np.abs(np.random.default_rng(H * 100 + W).standard_normal((n, 3))) + 0.1
    e = R.grid_edges(X, H, W, diag, nn)
    if e.shape[0] == 0:
        pytest.skip("no edges")
    cands = grid_candidates(n, e)
    assert len(cands) <= 8
    # a geometry that generates exactly this edge set is offered (the true one, or an equivalent: a 1 x W block has no
    # diagonal edges, so its 4- and 8-neighbour stencils coincide; an H x 1 column is the same chain as a 1 x H row)
    edges_of = lambda c: R.grid_edges(X, c[0], c[1], c[2], c[3])[:, :2]
    assert any(np.array_equal(edges_of(c), e[:, :2]) for c in cands), (cands, (H, W, diag, nn))


def test_grid_candidates_reject_a_general_graph():
    from phylo_hmrf_amd.pygco_compat import grid_candidates
    e = np.array([[0, 5], [0, 9], [1, 2], [3, 7]])
    # whatever is proposed must still pass the library's edge-by-edge check; here the node count (11) fits no triangle
    # and no width derived from node 0's neighbours divides it
    assert grid_candidates(11, e) == []
    assert grid_candidates(4, np.zeros((0, 2))) == []


# ---- workloads and sharding (bench.py --gpus N, phyloHMRF with world > 1) ----------------------------------------
def test_workload_block_lists():
    from phylo_hmrf_amd import workloads as W
    b3, S, K, nn, _ = W.workload("cfg3")
    assert (S, K, nn) == (4, 20, 8) and len(b3) == 26                 # 22 autosomes, chr3 / chr6 split in three
    n3 = sum(W.block_nodes(*b) for b in b3)
    assert n3 == 88833531 and max(W.block_nodes(*b) for b in b3) == 12402690          # chr1: 4980 bins at 50 kb
    assert sum(1 for b in b3 if not b[2]) == 2                        # the two off-diagonal blocks
    b5, S5, K5, _, _ = W.workload("cfg5")
    n5 = sum(W.block_nodes(*b) for b in b5)
    assert (S5, K5) == (4, 20) and len(b5) == 26 and 2.2e9 < n5 < 2.24e9              # <= 2.23 G nodes
    assert max(W.block_nodes(*b) for b in b5) == 24896 * 24897 // 2 < 2 ** 31 - 64    # fits the int32 node ids
    assert W.workload("cfg5-chr1")[0] == [(24896, 24896, True)]
    assert W.workload("cfg2")[0] == [(2000, 2000, True)] and W.workload("cfg4")[1:3] == (8, 30)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_shard_deals_every_block_once_and_balances(world, scaling):
    from phylo_hmrf_amd import workloads as W
    blocks = W.workload("cfg3")[0]
    all_blocks, owner = W.shard(blocks, world, scaling)
    sizes = np.array([W.block_nodes(*b) for b in all_blocks])
    assert len(all_blocks) == len(blocks) * (world if scaling == "weak" else 1)
    assert set(owner.tolist()) <= set(range(world)) and len(owner) == len(all_blocks)
    loads = np.array([sizes[owner == r].sum() for r in range(world)])
    assert loads.sum() == sizes.sum()
    # longest-processing-time-first: the fullest rank is within 4/3 - 1/(3 world) of the optimum, and the optimum is at
    # least max(mean load, largest block)
    lower = max(sizes.sum() / float(world), sizes.max())
    assert loads.max() <= (4.0 / 3.0 - 1.0 / (3.0 * world)) * lower + 1
    if scaling == "strong" and world == 8:
        assert loads.max() == sizes.max()                              # the chr1 block alone bounds the speed-up: 7.16x
    # deterministic: every rank computes the same deal
    assert np.array_equal(owner, W.shard(blocks, world, scaling)[1])


# ---- row tiles: the planner and the tile geometry (phylo_hmrf_amd/tiles.py; the solves themselves: tests/test_gpu_tiles.py) ----
@pytest.mark.parametrize("H,W,diag,parts", [(4980, 4980, True, 2), (4980, 4980, True, 3), (4980, 4980, True, 7), (1806, 2090, False, 3),
                                            (100, 100, True, 5), (64, 40, False, 2), (17, 17, True, 9)])
def test_split_rows_partitions_the_rows_into_balanced_tiles(H, W, diag, parts):
    from phylo_hmrf_amd import tiles
    rows = tiles.split_rows(H, W, diag, parts)
    assert rows[0][0] == 0 and rows[-1][1] == H
    for (a0, a1), (b0, b1) in zip(rows, rows[1:]):
        assert a1 == b0
    assert all(r1 - r0 >= tiles.MIN_TILE_ROWS for r0, r1 in rows) or len(rows) == 1
    assert len(rows) == min(parts, max(1, H // tiles.MIN_TILE_ROWS))
    n = [tiles.rows_nodes(r0, r1, W, diag) for r0, r1 in rows]
    assert sum(n) == (H * W - H * (H - 1) // 2 if diag else H * W)
    if H >= 20 * len(rows):          # equal shares up to a row or two of the longest row
        assert max(n) - min(n) <= 2 * W + 2 * tiles.MIN_TILE_ROWS * W * (H < 200)


def test_plan_cuts_the_blocks_above_a_ranks_share_and_balances_eight_ranks():
    from phylo_hmrf_amd import tiles, workloads
    blocks = workloads.genome_blocks(50000)
    total = sum(workloads.block_nodes(*b) for b in blocks)
    for world in (1, 2, 4):
        units = tiles.plan(blocks, world)
        assert len(units) == len(blocks) and all(u["ntiles"] == 1 for u in units)       # nothing above a rank's share
    units = tiles.plan(blocks, 8)
    split = sorted(set(u["block"] for u in units if u["ntiles"] > 1))
    assert split == [0, 1]                                                               # chr1 and chr2 (12.4 M, 11.7 M > 11.1 M)
    assert sum(u["nodes"] for u in units) == total
    owner = tiles.assign(units, 8)
    load = [sum(u["nodes"] for u, o in zip(units, owner) if o == r) for r in range(8)]
    assert max(load) <= 1.03 * total / 8.0, load                                         # whole blocks alone: 1.12 (chr1)
    # every unit of a split block names its rows; together they cover the block
    for b in split:
        us = [u for u in units if u["block"] == b]
        assert [u["tile"] for u in us] == list(range(len(us))) and us[0]["r0"] == 0 and us[-1]["r1"] == blocks[b][0]


@pytest.mark.parametrize("diag", [True, False])
def test_tile_geometry_is_the_blocks_rows(diag):
    from phylo_hmrf_amd import tiles
    H, W = (60, 60) if diag else (50, 70)

    class NoBlock(object):
        def __init__(self, n, S, K):
            self.n = n

    rows = tiles.split_rows(H, W, diag, 3)
    covered = 0
    for t, (r0, r1) in enumerate(rows):
        tl = tiles.Tile(H, W, diag, r0, r1, t, 3, 4, 5, NoBlock)
        assert (tl.top, tl.bottom) == (t > 0, t < 2)
        assert tl.Hl == (r1 - r0) + int(tl.top) + int(tl.bottom)
        assert tl.Wl == (W - tl.s0 if diag else W)
        assert tl.b.n == tl.n == (tl.Hl * tl.Wl - tl.Hl * (tl.Hl - 1) // 2 if diag else tl.Hl * tl.Wl)       # rows of a triangle are a smaller triangle's first rows
        assert tl.node0 == tiles.row_first(tl.s0, W, diag)
        assert tl.own_hi - tl.own_lo == tiles.rows_nodes(r0, r1, W, diag)
        g = tl.owned_global_slice()
        assert g.start == covered
        covered = g.stop
        # the rows that travel: what I send up is what the tile above receives from below, and so on
        if t > 0:
            up = tiles.Tile(H, W, diag, rows[t - 1][0], rows[t - 1][1], t - 1, 3, 4, 5, NoBlock)
            assert tl.top_out == up.bot_in and tl.top_in == up.bot_out
    assert covered == tiles.rows_nodes(0, H, W, diag)


def test_tile_group_payload_round_trip():
    from phylo_hmrf_amd import tiles
    g = tiles.TileGroup(0, (100, 100, True), 3, {}, None)
    buf = np.zeros(3 * g.slot, dtype=np.int64)
    c = np.arange(128, dtype=np.uint64) * 3 + (1 << 40)
    top, bot = np.arange(97, dtype=np.uint8) % 20, (np.arange(60, dtype=np.uint8) * 7) % 20
    g._pack(buf, 1, c, [1.5, -2.25e9], top, bot)
    o = g.slot
    assert np.array_equal(buf[o:o + 128].view(np.uint64), c)
    assert buf[o + 128:o + 130].view(np.float64).tolist() == [1.5, -2.25e9]
    rt, rb = g._rows(buf, 1)
    assert np.array_equal(rt[:97], top) and np.array_equal(rb[:60], bot)
    assert not buf[:o].any() and not buf[2 * o:].any()                # the other tiles' slots stay zero: an all-reduce is a gather


@pytest.mark.parametrize("H,W,diag", [(37, 37, True), (20, 31, False), (2, 2, True)])
def test_edge_lists_are_checked_against_the_grid_before_a_block_is_cut(H, W, diag):
    """tiles.edges_fit_grid: the host-side form of phmrf_block_set_grid's 'edge list joins nodes that are not grid neighbours'
    (the stencil of utility.py:1899-1905), asked before a block is cut into row tiles: a block whose list fails stays whole."""
    from phylo_hmrf_amd import tiles
    from phylo_hmrf_amd.graph_host import grid_edges
    n = tiles.rows_nodes(0, H, W, diag)
    i, j = tiles.node_coords(np.arange(n), W, diag)
    ii, jj = np.triu_indices(H) if diag else np.divmod(np.arange(n), W)
    assert np.array_equal(i, ii) and np.array_equal(j, jj)
    X = np.random.default_rng(0).random((n, 3)) + 0.1
    for nn in (8, 4):
        assert tiles.edges_fit_grid(grid_edges(X, H, W, diag, nn)[:, :2], H, W, diag, nn)
    e = grid_edges(X, H, W, diag, 8)
    if n > 10:
        bad = e.copy()
        bad[0, 1] = n - 1                                             # joins the first node with the last
        assert not tiles.edges_fit_grid(bad[:, :2], H, W, diag, 8)
        assert not tiles.edges_fit_grid(e[:, :2], H, W, diag, 4)      # diagonal edges in a 4-neighbour block
        bad = e.copy()
        bad[1, 0] = n                                                 # a node id outside the block
        assert not tiles.edges_fit_grid(bad[:, :2], H, W, diag, 8)
    # the coordinates hold at the 10 kb chr1 block's size (310 M nodes: the float root is corrected)
    W2 = 24896
    n2 = W2 * (W2 + 1) // 2
    ids = np.array([0, W2 - 1, W2, n2 - 1, n2 - 2, n2 - 3, 12345678901 % n2])
    i, j = tiles.node_coords(ids, W2, True)
    assert np.array_equal(i * W2 - (i * (i - 1)) // 2 + (j - i), ids) and (i <= j).all() and (j < W2).all()


def test_tile_holders_that_disagree_on_the_schedule_fail_instead_of_hanging():
    """Every holder of a block's tiles decides on the same sums; the status each decided travels with the next round's
    payload, and a disagreement raises on every rank (instead of leaving the others in the next all-reduce)."""
    from phylo_hmrf_amd import tiles

    class B(object):
        def solve_round_collect(self):
            return np.zeros(128, dtype=np.uint64), np.zeros(2)

        def tile_get_boundary(self, a, b):
            return (np.zeros(a, dtype=np.uint8) if a else None), (np.zeros(b, dtype=np.uint8) if b else None)

        def tile_put_halo(self, a, b):
            pass

        def solve_round_decide(self, c, e):
            return 0

    class Comm(object):                       # the other rank's tile reports that it decided "converged" last round
        def allreduce_i64(self, buf):
            buf = buf.copy()
            buf[1 * g.slot + 128 + 2] = 1 + 1
            return buf

    tl = tiles.Tile(40, 40, True, 0, 20, 0, 2, 4, 5, lambda n, S, K: B())
    g = tiles.TileGroup(0, (40, 40, True), 2, {0: tl}, Comm())
    with pytest.raises(RuntimeError, match="disagree on the schedule"):
        g.finish_round()
