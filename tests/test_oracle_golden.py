"""CPU: the NumPy oracle (oracle/ref_numpy.py) against fixtures recorded from the reference's own
functions (tests/golden/make_golden.py), and against scipy for the sklearn-0.18 density."""
import os

import numpy as np
import pytest
from scipy.stats import multivariate_normal

from oracle import ref_numpy as R

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag", ["t4", "t8"])
def test_tree_tables(tag):
    g = np.load(os.path.join(G, "tree_tables.npz"))
    tt = R.TreeTables(g[tag + "_edge_list"])
    assert np.array_equal(tt.leaf_vec, g[tag + "_leaf_vec"])
    assert [(-1 if p == [] else p) for p in tt.parent_list] == list(g[tag + "_parent_list"])
    assert np.array_equal(np.array(tt.pair_list), g[tag + "_pair_list"])
    assert np.array_equal(tt.A2, g[tag + "_A2"])
    assert sorted(tt.leaf_list.keys()) == list(g[tag + "_leaf_list_keys"])
    assert [tt.leaf_list[k] for k in sorted(tt.leaf_list)] == list(g[tag + "_leaf_list_vals"])
    assert tt.n_params == int(g[tag + "_n_params"])


def test_tree_tables_example_input_values():
    # SURVEY.md 8c G1 [probe]: leaves [2,5,6,7], parents [-,0,1,1,3,4,4,3]
    tt = R.TreeTables([[0, 1], [1, 2], [1, 3], [3, 4], [4, 5], [4, 6], [3, 7]])
    assert list(tt.leaf_vec) == [2, 5, 6, 7]
    assert tt.parent_list[1:] == [0, 1, 1, 3, 4, 4, 3]
    assert tt.pair_list == [[2, 5, 1], [2, 6, 1], [2, 7, 1], [5, 6, 4], [5, 7, 3], [6, 7, 3]]
    assert tt.n_params == 23


@pytest.mark.parametrize("tag", ["t4", "t8"])
def test_ou_params_to_mean_cov(tag):
    g1 = np.load(os.path.join(G, "tree_tables.npz"))
    g = np.load(os.path.join(G, "ou_params.npz"))
    tt = R.TreeTables(g1[tag + "_edge_list"])
    means, covars = R.ou_params_to_means_covars(tt, g[tag + "_params"])
    np.testing.assert_allclose(means, g[tag + "_means"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(covars, g[tag + "_covars"], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("tag", ["s4", "s8"])
def test_emission_matches_reference_and_scipy(tag):
    g = np.load(os.path.join(G, "emission.npz"))
    X, mu, cov = g[tag + "_X"], g[tag + "_means"], g[tag + "_covars"]
    lp = R.log_multivariate_normal_density_full(X, mu, cov)
    np.testing.assert_allclose(lp, g[tag + "_logprob"], rtol=1e-12, atol=1e-12)
    for k in (0, 1, mu.shape[0] - 1):
        np.testing.assert_allclose(lp[:, k], multivariate_normal(mu[k], cov[k]).logpdf(X), rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("et", [0, 3])
def test_posteriors_costs_stats(et):
    g = np.load(os.path.join(G, "posteriors_et%d.npz" % et))
    w, eid = R.edge_weights_from_distance(g["edges"], float(g["beta1"]))
    np.testing.assert_allclose(w, g["w"], rtol=0, atol=0)
    K = g["logprob"].shape[1]
    V = R.potts_matrix(K, float(g["beta"]))
    post, pc, pcn, uc, c1 = R.compute_posteriors_graph(g["labels"], g["logprob"], eid, w, V, et)
    np.testing.assert_allclose(post, g["posteriors"], rtol=1e-12, atol=1e-15)
    for mine, key in ((pc, "pairwise_cost"), (pcn, "pairwise_cost_normalize"), (uc, "unary_cost"), (c1, "cost1")):
        np.testing.assert_allclose(mine, float(g[key]), rtol=1e-12)
    st = R.sufficient_statistics(post, g["X"])
    np.testing.assert_allclose(st["post"], g["post"], rtol=1e-12)
    np.testing.assert_allclose(st["obs"], g["obs"], rtol=1e-12)
    np.testing.assert_allclose(st["obs*obs.T"], g["obsobsT"], rtol=1e-12)
    # the literal loop forms agree with the vectorised forms (incl. the isolated last node)
    pp_v = R.pairwise_compare(g["labels"], eid, w, V, et)
    pp_l = R.pairwise_compare_loops(g["labels"], eid, w, V, et)
    np.testing.assert_allclose(pp_v, pp_l, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(R.pairwise_cost_ensemble_loops(g["labels"], eid, w, V, et), pc, rtol=1e-12)
    # relation of SURVEY.md 8(a-E): E_float = n*unary_cost + (n/2)*pairwise_cost when estimate_type == 3
    if et == 3:
        n = len(g["labels"])
        e, eu, ep = R.mrf_energy(g["labels"], g["logprob"], eid, w, float(g["beta"]))
        np.testing.assert_allclose(e, n * uc + 0.5 * n * pc, rtol=1e-12)


@pytest.mark.parametrize("tag", ["chain", "diag", "offdiag"])
def test_energy_function_on_gco_labels(tag):
    g = np.load(os.path.join(G, "gco_%s.npz" % tag))
    w, eid = R.edge_weights_from_distance(g["edges"], 0.5)
    np.testing.assert_allclose(w, g["w"])
    for alg in ("swap", "expansion"):
        for q in ("pygco", "fine"):
            e = R.mrf_energy(g["labels_%s_%s" % (alg, q)], g["logprob"], eid, w, float(g["beta"]))
            np.testing.assert_allclose(e, g["efloat_%s_%s" % (alg, q)], rtol=1e-12)
            assert e[0] < float(g["efloat_init"][0])


def test_grid_edges_shapes_and_order():
    rng = np.random.default_rng(0)
    N = 7
    X = rng.uniform(0.1, 2, (N * (N + 1) // 2, 3))
    e = R.grid_edges(X, N, N, True, 8)
    assert np.all(e[:, 0] < e[:, 1])
    assert np.all(np.lexsort((e[:, 1], e[:, 0])) == np.arange(len(e)))
    # upper-tri 8-nbr half stencil: right N(N-1)/2, lower-right N(N-1)/2, lower (N-1)N/2... counted directly
    ii, jj = np.triu_indices(N)
    cnt = 0
    for dx, dy in [(0, 1), (1, 1), (1, 0), (1, -1)]:
        x2, y2 = ii + dx, jj + dy
        cnt += int(np.sum((x2 <= y2) & (y2 < N)))
    assert len(e) == cnt
    e4 = R.grid_edges(rng.uniform(0.1, 2, (20, 3)), 4, 5, False, 4)
    assert len(e4) == 4 * 4 + 3 * 5
