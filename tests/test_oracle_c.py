"""CPU: the plain-C restatement (oracle/estep_oracle.c) against the golden fixtures and the compiled reference gco."""
import os

import numpy as np
import pytest

from oracle import estep_c, gco_ref, ref_numpy as R

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag", ["s4", "s8"])
def test_c_emission_matches_reference_fixture(tag):
    g = np.load(os.path.join(G, "emission.npz"))
    lp = estep_c.emission(g[tag + "_X"], g[tag + "_means"], g[tag + "_covars"])
    np.testing.assert_allclose(lp, g[tag + "_logprob"], rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("et", [0, 3])
def test_c_posterior_stats_match_reference_fixture(et):
    g = np.load(os.path.join(G, "posteriors_et%d.npz" % et))
    stats, costs, post = estep_c.posterior_stats(g["X"], g["logprob"], g["edges"], g["w"], g["labels"], float(g["beta"]), et)
    K, S = g["logprob"].shape[1], g["X"].shape[1]
    np.testing.assert_allclose(post, g["posteriors"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(stats[:K], g["post"], rtol=1e-12)
    np.testing.assert_allclose(stats[K:K + K * S].reshape(K, S), g["obs"], rtol=1e-12)
    np.testing.assert_allclose(stats[K + K * S:].reshape(K, S, S), g["obsobsT"], rtol=1e-12)
    ref = [float(g["pairwise_cost"]), float(g["pairwise_cost_normalize"]), float(g["unary_cost"]), float(g["cost1"])]
    np.testing.assert_allclose(costs, ref, rtol=1e-12)


@pytest.mark.parametrize("tag", ["chain", "diag", "offdiag"])
@pytest.mark.parametrize("quant", ["pygco", "fine"])
def test_c_swap_reaches_the_reference_gco_energy(tag, quant):
    """Same integer problem, same move order: every alpha-beta swap is an exact min cut, so the integer energy after
    the run matches gco's recorded one (labels may differ only where a minimum cut is not unique)."""
    g = np.load(os.path.join(G, "gco_%s.npz" % tag))
    K = int(g["K"])
    w, eid = R.edge_weights_from_distance(g["edges"], 0.5)
    V = R.potts_matrix(K, float(g["beta"]))
    u_i, w_i, v_i = gco_ref.quantise(w, -g["logprob"], V, quant)
    lab, e, cycles = estep_c.swap_int(eid, w_i, u_i, v_i, g["init"])
    e_ref = int(g["eint_swap_%s" % quant][1])
    assert cycles >= 1
    assert abs(e - e_ref) <= 2e-4 * abs(e_ref), (e, e_ref)
    if np.array_equal(lab, g["labels_swap_%s" % quant]):
        assert e == e_ref
    e_float = estep_c.energy(g["logprob"], eid, w, lab, float(g["beta"]))
    np.testing.assert_allclose(e_float, R.mrf_energy(lab, g["logprob"], eid, w, float(g["beta"]))[0], rtol=1e-12)


@pytest.mark.skipif(not gco_ref.available(), reason="oracle/_ref not built")
def test_c_single_swap_move_equals_gco_move():
    """One alpha_beta_swap is a unique-value min cut: energies after ONE cycle agree exactly with gco's."""
    rng = np.random.default_rng(0)
    n, K = 400, 4
    eid = np.stack([np.arange(n - 1), np.arange(1, n)], 1)
    eid = np.concatenate([eid, np.stack([np.arange(n - 20), np.arange(20, n)], 1)])
    w_i = rng.integers(0, 50, len(eid)).astype(np.intc)
    u_i = rng.integers(0, 200, (n, K)).astype(np.intc)
    v_i = (1 - np.eye(K)).astype(np.intc) * 7
    init = rng.integers(0, K, n)
    lab_ref, e_ref = gco_ref.cut_general_graph_int(eid, w_i, u_i, v_i, n_iter=1, algorithm="swap", init_labels=init,
                                                   return_energy=True)
    lab, e, _ = estep_c.swap_int(eid, w_i, u_i, v_i, init, max_cycles=1)
    assert abs(e - e_ref["after"]) <= 1e-3 * abs(e_ref["after"])
