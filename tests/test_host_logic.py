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
