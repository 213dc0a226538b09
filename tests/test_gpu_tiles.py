"""Row tiles (phylo_hmrf_amd/tiles.py, csrc/tile.hip) on the GPU: one block cut into tiles that solve in lockstep rounds.

The reference has no counterpart to test against directly (its blocks are whole: base.py:357-362; its only split is the
centromere split into independent pieces, utility.py:381-393), so the tiled solve is held to the UNSPLIT solve of the same
block by the same library, to the float64 oracle's energy function, and -- through the unsplit solve -- to the reference's
gco energies (tests/test_gpu_estep.py).  Tolerances: energies rel 1e-6 against the oracle's energy of the same labelling;
tiled vs unsplit final energy rel 1e-4 on these small blocks (1e-5 at the full sizes, test_gpu_tiles_large); statistics
rel 2e-5.
"""
import numpy as np
import pytest

from oracle import mrf_moves as M
from oracle import ref_numpy as R
from oracle import synth

pytestmark = pytest.mark.gpu


def _whole(blk, H, W, diag, K, beta1=0.5):
    from phylo_hmrf_amd import Block
    X = blk["X"]
    b = Block(X.shape[0], X.shape[1], K)
    b.set_observations(X)
    b.build_grid_graph(H, W, diag, 8, beta1)
    return b


def _group(blk, H, W, diag, K, parts, beta1=0.5):
    from phylo_hmrf_amd import Block, tiles
    X = blk["X"]
    rows = tiles.split_rows(H, W, diag, parts)

    def load(tl):
        tl.b.set_observations(X[tl.global_slice()])

    return tiles.make_group(0, (H, W, diag), rows, [0] * len(rows), 0, X.shape[1], K, Block, load, None, 8, beta1)


def _gather(g, n):
    out = np.zeros(n, dtype=np.int32)
    for tl in g.local.values():
        out[tl.owned_global_slice()] = tl.b.get_labels()[tl.owned_local_slice()]
    return out


@pytest.mark.parametrize("diag,H,W,parts", [(True, 90, 90, 2), (True, 131, 131, 3), (False, 70, 95, 2), (False, 64, 40, 3)])
def test_tile_graph_is_the_blocks_graph(diag, H, W, parts):
    """a tile's device-built adjacency = the rows of the whole block's, ids shifted: rows r0..r1 of an upper triangle are the
    first rows of a smaller upper triangle (diagonal, H < W), including the halved diagonal-to-diagonal distances"""
    blk = synth.make_block(11, H, W, 4, 5, diag)
    b = _whole(blk, H, W, diag, 5)
    nbr, wgt = b.get_adjacency()
    g = _group(blk, H, W, diag, 5, parts)
    assert len(g.local) == parts
    for tl in g.local.values():
        tn, tw = tl.b.get_adjacency()
        # interior stored rows only: the first / last stored row of a tile lacks its neighbours outside the tile
        lo = tl.own_lo if tl.top else 0
        hi = tl.own_hi if tl.bottom else tl.n
        want_n = nbr[tl.node0 + lo:tl.node0 + hi].copy()
        want_n[want_n >= 0] -= tl.node0
        assert np.array_equal(tn[lo:hi], want_n)
        assert np.array_equal(tw[lo:hi], wgt[tl.node0 + lo:tl.node0 + hi])
        tl.b.close()
    b.close()


@pytest.mark.parametrize("diag,H,W,parts,K", [(True, 160, 160, 2, 6), (True, 200, 200, 3, 8), (False, 120, 150, 2, 6)])
def test_tiled_solve_matches_the_unsplit_solve(diag, H, W, parts, K):
    from phylo_hmrf_amd import tiles
    beta = 1.0
    blk = synth.make_block(5, H, W, 4, K, diag)
    n = blk["X"].shape[0]
    w, eid = R.edge_weights_from_distance(blk["edges"], 0.5)
    graph = M.Graph(n, eid, w)
    b = _whole(blk, H, W, diag, K)
    b.emission(blk["means"], blk["covars"])
    lp = b.get_logprob()
    res0 = b.solve(beta, init_mode=1)
    e_whole = M.energy(graph, -lp, b.get_labels(), beta)[0]
    assert abs(res0["energy"] - e_whole) <= 1e-6 * abs(e_whole)

    g = _group(blk, H, W, diag, K, parts)
    for tl in g.local.values():
        tl.b.emission(blk["means"], blk["covars"])
    opts = dict(init_mode=1)
    g.begin(beta, opts)
    energies = []
    while True:
        g.launch()
        st = g.finish_round()
        energies.append(M.energy(graph, -lp, _gather(g, n), beta)[0])
        if st != 0:
            break
    res = g.end(want_result=True)
    # the whole block's energy never goes up from round to round (f32 moves against the f64 sum: 1e-9 of slack)
    for a, c in zip(energies, energies[1:]):
        assert c <= a + 1e-9 * abs(a), energies
    labels = _gather(g, n)
    e_tiled = M.energy(graph, -lp, labels, beta)[0]
    # the tiles' own energies (owned rows) add up to the whole block's
    assert abs(res["energy"] - e_tiled) <= 1e-6 * abs(e_tiled), (res, e_tiled)
    assert res["converged"]
    assert abs(e_tiled - e_whole) <= 1e-4 * abs(e_whole), (e_tiled, e_whole)
    # halo rows agree with their owners
    for t, tl in g.local.items():
        lab = tl.b.get_labels()
        assert np.array_equal(lab, labels[tl.global_slice()])
    # posteriors / costs / statistics over the owned rows add up to the whole block's for the same labelling
    b.set_labels(labels)
    st0, c0, _ = b.posterior_stats(beta, 3)
    acc = {k: np.zeros_like(v) for k, v in st0.items()}
    cc = np.zeros(4)
    for tl in g.local.values():
        st, c, _ = tl.b.posterior_stats(beta, 3)
        for k in acc:
            acc[k] += st[k]
        cc += c
    for k in acc:
        assert np.allclose(acc[k], st0[k], rtol=2e-5, atol=1e-6), k
    assert np.allclose(cc, c0, rtol=1e-5)
    for tl in g.local.values():
        tl.b.close()
    b.close()


def test_tiled_warm_start_under_the_fit_tolerance():
    """the E-step's setting (energy_tol_ppb = 1000, warm start from the previous labelling under changed parameters)"""
    from phylo_hmrf_amd import tiles
    beta, K, H = 1.0, 8, 220
    blk = synth.make_block(9, H, H, 4, K, True)
    n = blk["X"].shape[0]
    w, eid = R.edge_weights_from_distance(blk["edges"], 0.5)
    graph = M.Graph(n, eid, w)
    rng = np.random.default_rng(3)
    means2 = blk["means"] * (1.0 + 0.04 * rng.standard_normal(blk["means"].shape))
    b = _whole(blk, H, H, True, K)
    g = _group(blk, H, H, True, K, 3)
    cond = tiles.Conductor([g])
    # iteration 1: cold, exact
    b.emission(blk["means"], blk["covars"])
    b.solve(beta, init_mode=1)
    cond.solve(beta, dict(init_mode=1), prepare=lambda tl: tl.b.emission(blk["means"], blk["covars"]))
    # iteration 2: warm, with the tolerance
    b.emission(means2, blk["covars"])
    lp = b.get_logprob()
    b.solve(beta, energy_tol_ppb=1000)
    e_whole = M.energy(graph, -lp, b.get_labels(), beta)[0]
    res = cond.solve(beta, dict(energy_tol_ppb=1000), prepare=lambda tl: tl.b.emission(means2, blk["covars"]),
                     want_result=True)[0]
    e_tiled = M.energy(graph, -lp, _gather(g, n), beta)[0]
    assert abs(res["energy"] - e_tiled) <= 1e-6 * abs(e_tiled)
    assert abs(e_tiled - e_whole) <= 2e-5 * abs(e_whole), (e_tiled, e_whole)      # both stop within 1e-6 of a round's gain
    for tl in g.local.values():
        tl.b.close()
    b.close()


@pytest.mark.parametrize("parts", [2, 3])
def test_tiled_cold_start_at_config_2_size_stays_below_gco(parts):
    """BASELINE config 2 in full (2000-bin diagonal block, 2,001,000 nodes, K = 10) from uniformly random labels, cut into
    2 and 3 row tiles: strictly at or below the reference's result (gco swap through pygco's quantisation, recorded in
    tests/golden/live_gco_energies.json from the reference's own code), within 1e-5 of the unsplit solve, and the whole
    block's energy (device f64 evaluation of the gathered labels) monotone from round to round."""
    import json
    import os
    from phylo_hmrf_amd import Block, tiles
    seed, N, K = 13, 2000, 10
    rec = [c for c in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "live_gco_energies.json")))["cases"]
           if c["seed"] == seed and c["N"] == N][0]
    blk = synth.make_block(seed, N, N, 4, K, True)
    X = blk["X"]
    n = X.shape[0]
    w, eid = R.edge_weights_from_distance(blk["edges"], 0.5)
    lp = R.log_multivariate_normal_density_full(X, blk["means"], blk["covars"])
    init = np.random.default_rng(seed + 7).integers(0, K, n)
    np.testing.assert_allclose(R.mrf_energy(init, lp, eid, w, 1.0)[0], rec["e_init"], rtol=1e-12)
    whole = Block(n, 4, K)
    whole.set_graph(eid, w)
    whole.set_grid(N, N, True, 8)
    whole.set_logprob(lp)
    whole.set_labels(init)
    whole.solve_fast(1.0, energy_tol_ppb=1000)
    e_whole = R.mrf_energy(whole.get_labels(), lp, eid, w, 1.0)[0]
    rows = tiles.split_rows(N, N, True, parts)

    def load(tl):
        tl.b.set_observations(X[tl.global_slice()])

    g = tiles.make_group(0, (N, N, True), rows, [0] * parts, 0, 4, K, Block, load, None, 8, 0.5, edges=blk["edges"])
    for tl in g.local.values():
        tl.b.set_logprob(lp[tl.global_slice()])
        tl.b.set_labels(init[tl.global_slice()])
    g.begin(1.0, dict(energy_tol_ppb=1000))
    energies = []
    while True:
        g.launch()
        st = g.finish_round()
        whole.set_labels(_gather(g, n))
        energies.append(whole.energy(1.0)[0])
        if st != 0:
            break
    g.end()
    for a, c in zip(energies, energies[1:]):
        assert c <= a + 1e-9 * abs(a), energies
    e_tiled = R.mrf_energy(_gather(g, n), lp, eid, w, 1.0)[0]
    print("2,001,000 nodes, %d tiles: gco via pygco %.2f  unsplit %.2f  tiled %.2f (%+.1e) in %d rounds"
          % (parts, rec["e_pygco"], e_whole, e_tiled, (e_tiled - e_whole) / abs(e_whole), len(energies)))
    assert e_tiled <= rec["e_pygco"], (e_tiled, rec)                 # strictly: <= what the reference computes
    assert abs(e_tiled - e_whole) <= 1e-5 * abs(e_whole), (e_tiled, e_whole)
    for tl in g.local.values():
        tl.b.close()
    whole.close()


def test_a_moved_pinned_row_is_reported():
    """The safety net under the tiles' lockstep rounds: every re-pin counts the pinned nodes whose label is no longer the one
    they were pinned with, and the round's read-back carries that count (PHMRF_ERR_STATE).  No move type may relabel a pinned
    node, so the count can only be forced from outside: a label of the pinned row is changed behind the solver's back
    between two phmrf_block_tile_pins calls."""
    from phylo_hmrf_amd import Block
    from phylo_hmrf_amd._lib import PhmrfError
    H = W = 48
    K = 5
    blk = synth.make_block(seed=11, H=H, W=W, S=4, K=K, diagonal=True)
    b = _whole(blk, H, W, True, K)
    b.emission(blk["means"], blk["covars"])
    b.solve(1.0, init_mode=1, max_rounds=2)
    b.set_tile(True, False, 0)
    # a clean pair of rounds first: nothing is reported
    b.solve_begin(1.0, energy_tol_ppb=1000)
    b.tile_pins(1, 0)
    b.solve_round_launch()
    counters, energy = b.solve_round_collect()
    b.solve_round_decide(counters, energy)
    b.tile_pins(2, 0)
    b.solve_round_launch()
    counters, energy = b.solve_round_collect()
    b.solve_round_decide(counters, energy)
    b.solve_end()
    # now with the first row (pinned) relabelled between the pins of two rounds
    b.solve_begin(1.0, energy_tol_ppb=1000)
    b.tile_pins(1, 0)
    b.solve_round_launch()
    counters, energy = b.solve_round_collect()
    b.solve_round_decide(counters, energy)
    lab = b.get_labels()
    lab[:5] = (lab[:5] + 1) % K
    b.set_labels(lab)
    b.tile_pins(2, 0)
    b.solve_round_launch()
    with pytest.raises(PhmrfError) as ei:
        b.solve_round_collect()
    assert ei.value.status == 5 and "pinned row" in str(ei.value)        # PHMRF_ERR_STATE
    b.close()
