"""CPU: the oracle on REAL Hi-C -- oracle/ref_numpy.py against what the reference's own fit recorded on the example's
chr22 block (tests/golden/example_chr22_em.npz, written by tests/golden/make_golden_example.py)."""
import os

import numpy as np
import pytest

from oracle import ref_numpy as R

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ex():
    g = np.load(os.path.join(G, "example_chr22_em.npz"))
    return {k: g[k] for k in g.files}


def test_fixture_is_the_example_block(ex):
    lv = ex["len_vec"][0]
    n = ex["X"].shape[0]
    assert lv[0] == n == lv[3] * (lv[3] + 1) // 2 and lv[8] == 1 and lv[9] == 22       # diagonal block of chr22
    assert ex["it_labels"].shape == (5, n) and ex["cost_vec"].shape == (5, 4)
    # the edge list is the reference's stencil on this block, and the oracle's builder reproduces it bit for bit
    e = R.grid_edges(ex["X"], int(lv[3]), int(lv[4]), True, 8)
    assert np.array_equal(e, ex["edges"])


@pytest.mark.parametrize("it", [0, 4])
def test_oracle_costs_and_stats_on_the_reference_labels(ex, it):
    beta = float(ex["beta"])
    K = int(ex["K"])
    w, eid = R.edge_weights_from_distance(ex["edges"], float(ex["beta1"]))
    lp = R.log_multivariate_normal_density_full(ex["X"], ex["it_means"][it], ex["it_covars"][it])
    lab = np.int64(ex["it_labels"][it])
    e = R.mrf_energy(lab, lp, eid, w, beta)
    np.testing.assert_allclose(e, ex["it_efloat"][it], rtol=1e-10)
    post, pc, pcn, uc, c1 = R.compute_posteriors_graph(lab, lp, eid, w, R.potts_matrix(K, beta), 3)
    np.testing.assert_allclose([pcn, uc, c1], ex["cost_vec"][it][1:4], rtol=1e-9)
    st = R.sufficient_statistics(post, ex["X"])
    np.testing.assert_allclose(st["post"], ex["it_stats_post"][it], rtol=1e-9)
    np.testing.assert_allclose(st["obs*obs.T"], ex["it_stats_oo"][it], rtol=1e-9)


def test_reference_labelling_can_raise_the_float_energy_on_real_data(ex):
    """What 'E <= the reference's' means on real Hi-C: pygco scales every term by max|unary| before truncating to
    integers, so with the large |logprob| of real data most edge weights become 0 and gco's swap, which
    lowers the INTEGER energy, can return labels whose float energy is above the warm start's."""
    d = ex["it_efloat"][:, 0] - ex["it_efloat_init"][:, 0]
    assert (d > 0).any() and (d < 0).any()


def test_oracle_energy_on_the_full_chr22_block():
    """The full 683-bin chr22 block (tests/golden/example_chr22_full.npz, BASELINE config 1's block size): the oracle's
    edge builder gives the edge count the reference was handed, and the oracle's float64 energy of the reference's own
    labels is what the generator recorded (two of the five iterations; each is a 233,586 x 20 density evaluation)."""
    g = np.load(os.path.join(G, "example_chr22_full.npz"))
    lv = g["len_vec"][0]
    X = np.float64(g["X"])
    n = X.shape[0]
    assert lv[0] == n == 683 * 684 // 2 and lv[3] == lv[4] == 683 and lv[8] == 1 and lv[9] == 22
    edges = R.grid_edges(X, 683, 683, True, 8)
    assert edges.shape[0] == int(g["n_edges"])
    w, eid = R.edge_weights_from_distance(edges, float(g["beta1"]))
    for it in (1, 4):
        lp = R.log_multivariate_normal_density_full(X, g["it_means"][it], g["it_covars"][it])
        e = R.mrf_energy(np.int64(g["it_labels"][it]), lp, eid, w, float(g["beta"]))
        np.testing.assert_allclose(e, g["it_efloat"][it], rtol=1e-10)
        e0 = R.mrf_energy(np.int64(g["it_init"][it]), lp, eid, w, float(g["beta"]))
        np.testing.assert_allclose(e0, g["it_efloat_init"][it], rtol=1e-10)
    # the reference's labelling of iteration 1 has a HIGHER float energy than its warm start (pygco's quantisation)
    assert g["it_efloat"][1][0] > g["it_efloat_init"][1][0]
