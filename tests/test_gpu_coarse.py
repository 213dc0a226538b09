"""GPU: the coarse alpha-expansion (phylo_hmrf_amd/csrc/coarse.hip) against its float64 model
(oracle/mrf_moves.coarse_problem / coarse_expansion), and what it is for: regions taller than a strip."""
import numpy as np
import pytest

from oracle import mrf_moves as M
from oracle import ref_numpy as R
from oracle import synth
from tests.test_gpu_estep import _block, _integer_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,K,diagonal", [(41, 41, 5, True), (26, 70, 6, False), (9, 9, 3, True), (4, 5, 2, False)])
@pytest.mark.parametrize("scale", [2, 4, 8])
def test_coarse_problem_matches_model(H, W, K, diagonal, scale):
    n, eid, w, lp, init = _integer_problem(9, H, W, K, diagonal)
    g = M.Graph(n, eid, w)
    b = _block(n, 2, K)
    b.set_graph(eid, w)
    b.set_grid(H, W, diagonal, 8)
    b.set_logprob(lp)
    b.set_labels(init)
    for off in range(scale):
        alpha = (off + 1) % K
        D_ref, lam_ref, cnode, Hc, Wc = M.coarse_problem(g, -lp, init, 0.75, H, W, diagonal, scale, off, alpha)
        D, lam = b.coarse_problem(0.75, scale, off, alpha)
        assert D.shape == D_ref.shape and lam.shape == lam_ref.shape
        # a super-cell that costs more than all its cross edges weigh is pinned (D = 1e30): it is in no optimal switch set
        pinned = D_ref > 0.75 * M.coarse_cross_weight(g, H, W, diagonal, scale, off) * 1.0001 + 1e-6
        assert np.array_equal(D > 1e29, pinned)
        assert np.array_equal(D[~pinned].astype(np.float64), D_ref[~pinned])      # dyadic inputs: f32 sums are exact
        assert np.array_equal(lam.astype(np.float64), lam_ref)
        assert lam.min() >= 0.0                                         # Potts expansions are submodular
    b.close()


@pytest.mark.parametrize("H,W,K,diagonal", [(60, 60, 5, True), (33, 90, 4, False), (150, 150, 8, True)])
def test_coarse_pass_matches_move_model(H, W, K, diagonal):
    n, eid, w, lp, init = _integer_problem(21, H, W, K, diagonal)
    g = M.Graph(n, eid, w)
    b = _block(n, 2, K)
    b.set_graph(eid, w)
    b.set_grid(H, W, diagonal, 8)
    b.set_logprob(lp)
    b.set_labels(init)
    lab = init.astype(np.int64).copy()
    for it, (scale, off, alpha, sr, sc) in enumerate([(2, 0, 1, 0, 0), (4, 3, 0, 2, 40), (2, 1, 2, 5, 63), (4, 0, 3, 1, 7),
                                                       (2, 0, 0, 3, 21)]):
        e0 = M.energy(g, -lp, lab, 1.0)[0]
        ch_ref = M.coarse_expansion(g, -lp, lab, 1.0, H, W, diagonal, scale, off, alpha, sr, sc)
        ch = b.coarse_pass(1.0, scale, off, alpha, sr, sc)
        got = b.get_labels().astype(np.int64)
        assert M.energy(g, -lp, lab, 1.0)[0] <= e0 + 1e-9
        assert np.array_equal(got, lab), (it, int((got != lab).sum()))
        assert ch == ch_ref
    b.close()


def test_coarse_moves_reach_regions_taller_than_a_strip():
    """Two nearly identical states; a 24 x 30 rectangle of one inside the other, labelled like its surroundings.  Every
    5-row slice of the rectangle loses (two long new boundaries for a thin gain); the rectangle as a whole wins.  The
    fine moves leave it standing, the coarse expansion takes it."""
    H = W = 80
    n = H * W
    K = 3
    rng = np.random.default_rng(2)
    X = rng.uniform(0.5, 2.0, (n, 2))
    e = R.grid_edges(X, H, W, False, 8)
    eid = np.int64(e[:, :2])
    w = np.full(len(eid), 0.75)
    truth = np.zeros((H, W), dtype=np.int64)
    truth[28:52, 25:55] = 1
    un = np.full((n, K), 4.0)
    un[np.arange(n), truth.reshape(-1)] = 0.0
    un[truth.reshape(-1) == 1, 0] = 0.5             # inside the rectangle label 0 is only slightly worse than label 1:
    un[truth.reshape(-1) == 0, 1] = 0.5             # 360 in all, against 240 of new boundary -- but a 5-row slice gains
    #                                                 75 and pays 150, a 10-row slab gains 150 and pays 180; 20 rows win
    lp = -un
    g = M.Graph(n, eid, w)
    b = _block(n, 2, K)
    b.set_graph(eid, w)
    b.set_grid(H, W, False, 8)
    b.set_logprob(lp)
    b.set_labels(np.zeros(n, dtype=np.int64))
    res_fine = b.solve(1.0, use_coarse=False)
    assert res_fine["changed"] == 0                                  # a fixed point of every fine move
    b.set_labels(np.zeros(n, dtype=np.int64))
    res = b.solve(1.0, use_coarse=True)
    lab = b.get_labels()
    e_truth = M.energy(g, un, truth.reshape(-1), 1.0)[0]
    assert res["energy"] < res_fine["energy"] - 50.0
    # the rectangle is found (with its corners rounded off: cheaper than the exact rectangle)
    assert res["energy"] <= e_truth + 1e-9 and np.mean(lab == truth.reshape(-1)) > 0.995 and (lab == 2).sum() == 0
    b.close()


def test_coarse_passes_never_raise_the_energy():
    blk = synth.make_block(5, 120, 120, 4, 10, True)
    X = blk["X"]
    n = X.shape[0]
    w, eid = R.edge_weights_from_distance(blk["edges"], 0.5)
    b = _block(n, 4, 10)
    b.set_observations(X)
    b.set_graph(eid, w)
    b.set_grid(120, 120, True, 8)
    b.emission(blk["means"], blk["covars"])
    b.set_labels(np.random.default_rng(0).integers(0, 10, n))
    prev = b.energy(1.0)[0]
    rng = np.random.default_rng(1)
    total = 0
    for it in range(30):
        scale = 2 if it % 2 == 0 else 4
        total += b.coarse_pass(1.0, scale, int(rng.integers(0, scale)), it % 10, int(rng.integers(0, 6)), int(rng.integers(0, 64)))
        e = b.energy(1.0)[0]
        assert e <= prev + 1e-6 * abs(prev), (it, e, prev)
        prev = e
    assert total > 0
    b.close()


def test_coarse_to_fine_start_is_a_start_and_nothing_else():
    """phmrf_solve_opts.coarse_start = 1 (c2f.hip): the cold start of a grid block from the solution of its 4 x 4 super-cell
    problem -- a Potts problem on the coarse grid whose unary terms and pair weights are the sums over the cells and over the
    crossing edges.  Only the START changes: the solve's
    energy never exceeds the start's, and the result stays within 1e-3 of the default cold start's (measured on the
    synthetic workloads: +8e-5 ... -3e-5; it is slower there, which is why the option is off by default).  The reference has
    no counterpart (gco's swap starts from the labels it is given, GCoptimization.cpp:1282-1305)."""
    from phylo_hmrf_amd import Block
    from oracle import mrf_moves as M
    from oracle import ref_numpy as R
    H = W = 320          # 51,360 nodes: the 4 x 4 child has 3,240 and starts coarse-to-fine itself (its own child: 210)
    K = 6
    blk = synth.make_block(seed=5, H=H, W=W, S=4, K=K, diagonal=True)
    X = blk["X"]
    n = X.shape[0]
    b = Block(n, 4, K)
    b.set_observations(X)
    b.build_grid_graph(H, W, True, 8, 0.5)
    b.emission(blk["means"], blk["covars"])
    lp = b.get_logprob()
    w, eid = R.edge_weights_from_distance(blk["edges"], 0.5)
    ref = b.solve(1.0, init_mode=1, energy_tol_ppb=1000)
    res = b.solve(1.0, init_mode=1, energy_tol_ppb=1000, coarse_start=1)
    lab = b.get_labels()
    e = R.mrf_energy(lab, lp, eid, w, 1.0)[0]
    assert abs(e - res["energy"]) <= 1e-6 * abs(e)
    assert res["energy"] <= res["energy_init"] + 1e-9 * abs(res["energy_init"])
    assert abs(res["energy"] - ref["energy"]) <= 1e-3 * abs(ref["energy"]), (res, ref)
    # the start is constant on the 4 x 4 super-cells: a solve that is allowed no round returns it
    b.solve_fast(1.0, init_mode=1, max_rounds=1, use_chains=False, use_components=False, use_expansion=False, coarse_start=1,
                 use_coarse=False, energy_tol_ppb=1000)
    b.close()
    b = Block(n, 4, K)
    b.set_observations(X)
    b.build_grid_graph(H, W, True, 8, 0.5)
    b.set_logprob(lp)
    # (an energy tolerance so large that the first round's result is accepted: what is looked at is energy_init)
    res0 = b.solve(1.0, init_mode=1, max_rounds=1, coarse_start=1)
    start_e = res0["energy_init"]
    assert np.isfinite(start_e) and start_e >= res0["energy"]
    b.close()
